// Which memory groups does an MI355X have?  66 hipMalloc'ed granules of 4 GiB; for several reference granules R the gather of
// tools/experiments/hbm_map.hip is run with its index and output streams placed INSIDE granule R and its random reads in every other granule:
// a low rate = "shares a group with R".  Prints one row per reference granule (rate per granule, '-' = low class).
//    hipcc -O3 --offload-arch=gfx950 -o build/hbm_groups tools/experiments/hbm_groups.hip ; build/hbm_groups [granules]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void gather8(const u64 *base, const u64 *idx, u64 *out, u64 n, u64 nlines)
{
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) out[i] = base[(idx[i] % nlines) * 8];
}
__global__ void fill_idx(u64 *idx, u64 n)
{
    const u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x;
    if (i < n) { u64 s = (i + 1) * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32; idx[i] = s >> 8; }
}
int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? atoi(argv[1]) : 66, gran = 4ull << 30;
    const u64 N = 1ull << 24;
    CK(hipSetDevice(0));
    std::vector<char *> g;
    for (size_t k = 0; k < n; k++) { void *p; if (hipMalloc(&p, gran) != hipSuccess) { (void)hipGetLastError(); break; } g.push_back((char *)p); }
    printf("%zu granules of 4 GiB\n", g.size());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<std::vector<float>> M;
    std::vector<size_t> refs;
    for (size_t r = 0; r < g.size(); r += 3) refs.push_back(r);
    for (size_t r : refs) {
        u64 *idx = (u64 *)g[r], *out = (u64 *)(g[r] + (1ull << 30));           // the streams live inside granule r
        hipLaunchKernelGGL(fill_idx, dim3(N / 256), dim3(256), 0, 0, idx, N);
        std::vector<float> row(g.size(), 0.f);
        printf("ref %2zu:", r);
        for (size_t k = 0; k < g.size(); k++) {
            if (k == r) { printf("  ## "); continue; }
            float ms = 0;
            hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const u64 *)g[k], idx, out, N, gran / 64);
            CK(hipEventRecord(a));
            for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const u64 *)g[k], idx, out, N, gran / 64);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b));
            row[k] = 3.0f * N / (ms * 1e6f);
            printf(" %4.1f", row[k]);
        }
        printf("\n");
        M.push_back(row);
    }
    // compact picture: per reference granule, which granules are in the low class (< 40.2)
    for (size_t i = 0; i < refs.size(); i++) {
        printf("ref %2zu: ", refs[i]);
        for (size_t k = 0; k < g.size(); k++) putchar(k == refs[i] ? '#' : M[i][k] < 40.2f ? '-' : M[i][k] < 41.6f ? 'o' : '+');
        printf("\n");
    }
    return 0;
}
