// vm_hint.hip -- does hipMemAddressReserve honour its address hint, and does a reservation freed and made again come back at the same address?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t chunk = 4ull << 30;
    void *a = nullptr, *b = nullptr, *c = nullptr, *d = nullptr, *e = nullptr;
    CK(hipMemAddressReserve(&a, chunk, 0, nullptr, 0)); printf("a (no hint)            %p\n", a);
    CK(hipMemAddressFree(a, chunk));
    CK(hipMemAddressReserve(&b, chunk, 0, nullptr, 0)); printf("b (no hint, a freed)   %p  %s\n", b, a == b ? "SAME as a" : "different");
    void *hint = (char *)b + (1ull << 40);
    CK(hipMemAddressReserve(&c, chunk, 0, hint, 0)); printf("c (hint b + 1 TiB)     %p  hint %p  %s\n", c, hint, c == hint ? "HONOURED" : "ignored");
    hint = (char *)hint + chunk;
    CK(hipMemAddressReserve(&d, 64 * chunk, 0, hint, 0)); printf("d (256 GiB, hint next) %p  hint %p  %s\n", d, hint, d == hint ? "HONOURED" : "ignored");
    hint = (void *)0x600000000000ull;
    CK(hipMemAddressReserve(&e, chunk, 0, hint, 0)); printf("e (hint 0x6000...)     %p  %s\n", e, e == hint ? "HONOURED" : "ignored");
    return 0;
}
