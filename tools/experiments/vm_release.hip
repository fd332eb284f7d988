// vm_release.hip -- how soon does memory taken with hipMemCreate come back after hipMemUnmap + hipMemRelease, as hipMemGetInfo and hipMalloc see it?
// (round 4: the chunk-composed bucket lines of placement.hip park what they do not need; a second engine on the same GPU must be able to get it)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(unsigned long long *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i * 512] = i; }
int main(int argc, char **argv)
{
    const int nchunks = argc > 1 ? atoi(argv[1]) : 24;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;          // 0: one arena, never freed; 1: one reservation per chunk, freed after the unmap; 2: arena, release BEFORE unmap
    printf("mode %d\n", mode);
    const size_t chunk = 4ull << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    void *arena = nullptr;
    std::vector<void *> va(nchunks);
    if (mode != 1) { CK(hipMemAddressReserve(&arena, (size_t)nchunks * chunk * 2, 0, nullptr, 0)); for (int k = 0; k < nchunks; k++) va[k] = va[k]; }
    else for (int k = 0; k < nchunks; k++) CK(hipMemAddressReserve(&va[k], chunk, 0, nullptr, 0));
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot)); printf("start: %.1f GiB free\n", fr / 1073741824.0);
    std::vector<hipMemGenericAllocationHandle_t> h(nchunks);
    for (int k = 0; k < nchunks; k++) {
        CK(hipMemCreate(&h[k], chunk, &prop, 0));
        CK(hipMemMap(va[k], chunk, 0, h[k], 0));
        CK(hipMemSetAccess(va[k], chunk, &acc, 1));
        hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (unsigned long long *)(va[k]), chunk / 4096);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemGetInfo(&fr, &tot)); printf("after taking %d chunks: %.1f GiB free\n", nchunks, fr / 1073741824.0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < nchunks; k++) {
        if (mode == 2) { CK(hipMemRelease(h[k])); CK(hipMemUnmap(va[k], chunk)); }
        else { CK(hipMemUnmap(va[k], chunk)); CK(hipMemRelease(h[k])); }
        if (mode == 1) CK(hipMemAddressFree(va[k], chunk));
    }
    for (int i = 0; i < 16; i++) {
        CK(hipMemGetInfo(&fr, &tot));
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, (size_t)(nchunks - 2) * chunk);
        printf("%.2f s after release: %.1f GiB free, hipMalloc of %d GiB: %s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), fr / 1073741824.0,
               (nchunks - 2) * 4, hipGetErrorString(e));
        if (e == hipSuccess) { (void)hipFree(p); break; }
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::milliseconds(250));
    }
    return 0;
}
