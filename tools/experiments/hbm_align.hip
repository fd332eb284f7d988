// Does the random-access speed of a big buffer depend on how its virtual address is aligned relative to its physical memory?
// (The page tables describe an aligned, physically contiguous range with one large "fragment"; a virtual address that is off by 2 MiB
// against the physical block forces 2 MiB fragments and the translation caches stop covering a 16 GiB table.)
// A 16 GiB buffer is built from physical granules (hipMemCreate) mapped at base + offset for several offsets, then read at random.
//    hipcc -O3 --offload-arch=gfx950 -o build/hbm_align tools/experiments/hbm_align.hip ; build/hbm_align [granule MiB] [total GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void random_lines(const v4u *base, unsigned long long nlines, unsigned reads, unsigned long long seed, unsigned *sink)
{
    unsigned long long s = seed + (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned acc = 0;
    for (unsigned r = 0; r < reads; r += 4) {
        v4u v[4];
        for (int k = 0; k < 4; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            v[k] = __builtin_nontemporal_load(base + ((s >> 20) % nlines) * 4);
        }
        for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].w;
    }
    if (acc == 0x12345u) *sink = 1;
}
__global__ void block_streams(v4u *base, unsigned long long per_block, unsigned *sink)
{
    v4u *p = base + blockIdx.x * per_block + threadIdx.x;
    unsigned acc = 0;
    for (unsigned long long i = 0; i < per_block; i += blockDim.x) __builtin_nontemporal_store((v4u){(unsigned)i, 1u, 2u, 3u}, p + i);
    for (unsigned long long i = per_block; i >= blockDim.x; i -= blockDim.x) acc ^= __builtin_nontemporal_load(p + i - blockDim.x).x;
    if (acc == 0x12345u) *sink = 1;
}
__global__ void gather8(const unsigned long long *base, const unsigned long long *idx, unsigned long long *out, unsigned long long n, unsigned long long nlines)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = base[(idx[i] % nlines) * 8];
}
__global__ void fill_idx(unsigned long long *idx, unsigned long long n)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) { unsigned long long s = (i + 1) * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32; idx[i] = s >> 8; }
}
static unsigned long long *gidx, *gout;
static unsigned *sink;
static hipEvent_t ea, eb;
static void grade(const char *tag, void *va, size_t bytes)
{
    float ms_r = 0, ms_s = 0;
    const unsigned long long nlines = bytes / 64;
    hipLaunchKernelGGL(random_lines, dim3(8192), dim3(256), 0, 0, (const v4u *)va, nlines, 32u, 1ull, sink);
    CK(hipEventRecord(ea));
    for (int rep = 0; rep < 8; rep++) hipLaunchKernelGGL(random_lines, dim3(8192), dim3(256), 0, 0, (const v4u *)va, nlines, 32u, 77ull + rep, sink);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms_r, ea, eb));
    CK(hipEventRecord(ea));
    hipLaunchKernelGGL(block_streams, dim3(2048), dim3(256), 0, 0, (v4u *)va, bytes / 16 / 2048, sink);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms_s, ea, eb));
    float ms_g = 0;
    const unsigned long long N = 1ull << 26;
    hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const unsigned long long *)va, gidx, gout, N, nlines);
    CK(hipEventRecord(ea));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const unsigned long long *)va, gidx, gout, N, nlines);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms_g, ea, eb));
    printf("%-64s va %p  random 64-byte reads %6.2f G/s   gather8 %6.2f G/s   2048 block streams %6.0f GB/s\n", tag, va, 8.0 * 8192 * 256 * 32 / (ms_r * 1e-3) / 1e9,
           3.0 * N / (ms_g * 1e-3) / 1e9, 2.0 * bytes / (ms_s * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const size_t gran = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20, total = (size_t)(argc > 2 ? atoi(argv[2]) : 16) << 30;
    const size_t n = total / gran;
    CK(hipSetDevice(0));
    CK(hipMalloc(&sink, 4)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    CK(hipMalloc(&gidx, 8ull << 26)); CK(hipMalloc(&gout, 8ull << 26));
    hipLaunchKernelGGL(fill_idx, dim3((1u << 26) / 256), dim3(256), 0, 0, gidx, 1ull << 26);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    // plain hipMalloc for comparison, three times
    for (int k = 0; k < 3; k++) {
        void *p; CK(hipMalloc(&p, total));
        grade("hipMalloc", p, total);
        void *q; CK(hipMalloc(&q, (size_t)(k + 1) << 30));          // shift what the next one gets
        CK(hipFree(p)); CK(hipFree(q));
    }
    std::vector<hipMemGenericAllocationHandle_t> hs(n);
    for (size_t k = 0; k < n; k++) CK(hipMemCreate(&hs[k], gran, &prop, 0));
    const size_t offs[] = {0, 2ull << 20, 4ull << 20, 8ull << 20, 32ull << 20, 128ull << 20, 512ull << 20, 1ull << 30, 2ull << 30, 0};
    const size_t noffs = sizeof offs / sizeof offs[0], slot = total + (8ull << 30);
    const size_t span = slot * noffs + (4ull << 30);
    void *va0 = nullptr;
    CK(hipMemAddressReserve(&va0, span, 0, nullptr, 0));
    char *va = (char *)(((uintptr_t)va0 + (4ull << 30) - 1) & ~((4ull << 30) - 1));          // the reservation call ignores its alignment argument
    printf("%zu granules of %zu MiB; reserved %p, aligned by hand to %p\n", n, gran >> 20, va0, (void *)va);
    for (size_t o = 0; o < noffs; o++) {
        const size_t off = offs[o];
        char *base = va + o * slot + off;
        for (size_t k = 0; k < n; k++) CK(hipMemMap(base + k * gran, gran, 0, hs[k], 0));
        CK(hipMemSetAccess(base, total, &acc, 1));
        char tag[96];
        snprintf(tag, sizeof tag, "granules mapped at 4 GiB-aligned address + %zu MiB", off >> 20);
        grade(tag, base, total);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(base, total));
    }
    CK(hipMemAddressFree(va0, span));
    for (size_t k = 0; k < n; k++) CK(hipMemRelease(hs[k]));
    return 0;
}
