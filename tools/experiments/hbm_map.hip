// Map of the HBM by random-access speed.  Memory is taken in granules (hipMemCreate handles mapped one by one, or plain hipMalloc with
// `malloc` as the 4th argument), each granule is graded, and -- handles only -- buffers composed of the fastest / slowest granules are
// graded again.  Findings on MI355X (profiles/r02g_hbm_map_*.log): high-parallelism random reads (random_lines) see no difference between
// granules; a gather with one 8-byte load per thread next to two coalesced streams (gather8, what torch's x[idx] does) is 10 % slower in
// some 64+ GiB of every box's memory.
//    hipcc -O3 --offload-arch=gfx950 -o build/hbm_map tools/experiments/hbm_map.hip ; build/hbm_map [GiB per granule] [max granules] [GiB composite] [malloc]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__global__ void random_lines(const v4u *base, u64 nlines, unsigned reads, u64 seed, unsigned *sink)
{
    u64 s = seed + (blockIdx.x * (u64)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned acc = 0;
    for (unsigned r = 0; r < reads; r += 4) {
        v4u v[4];
        for (int k = 0; k < 4; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const v4u *q = base + ((s >> 20) % nlines) * 4;
            v[k] = NT ? __builtin_nontemporal_load(q) : *q;
        }
        for (int k = 0; k < 4; k++) acc ^= v[k].x ^ v[k].w;
    }
    if (acc == 0x12345u) *sink = 1;
}
// what torch's x.view(-1, 8)[idx, 0] does: one 8-byte load per thread at a row taken from an index array, one coalesced store
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
// 2048 blocks, each streaming its own contiguous part 16 bytes per lane: write then read back (the chain scratch's pattern)
__global__ void block_streams(v4u *base, u64 per_block, unsigned *sink)
{
    v4u *p = base + blockIdx.x * per_block + threadIdx.x;
    unsigned acc = 0;
    for (u64 i = 0; i < per_block; i += blockDim.x) __builtin_nontemporal_store((v4u){(unsigned)i, 1u, 2u, 3u}, p + i);
    for (u64 i = per_block; i >= blockDim.x; i -= blockDim.x) acc ^= __builtin_nontemporal_load(p + i - blockDim.x).x;
    if (acc == 0x12345u) *sink = 1;
}

static unsigned *sink;
static u64 *gidx, *gout;
static hipEvent_t ea, eb;
struct Grade { double nt, plain, gather, streams; };
static Grade grade(void *va, size_t bytes)
{
    Grade g;
    float ms = 0;
    const u64 nlines = bytes / 64, N = 1ull << 26;
    hipLaunchKernelGGL(random_lines<true>, dim3(4096), dim3(256), 0, 0, (const v4u *)va, nlines, 16u, 1ull, sink);
    CK(hipEventRecord(ea));
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(random_lines<true>, dim3(8192), dim3(256), 0, 0, (const v4u *)va, nlines, 32u, 77ull + rep, sink);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb));
    g.nt = 4.0 * 8192 * 256 * 32 / (ms * 1e-3) / 1e9;
    CK(hipEventRecord(ea));
    for (int rep = 0; rep < 4; rep++) hipLaunchKernelGGL(random_lines<false>, dim3(8192), dim3(256), 0, 0, (const v4u *)va, nlines, 32u, 177ull + rep, sink);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb));
    g.plain = 4.0 * 8192 * 256 * 32 / (ms * 1e-3) / 1e9;
    hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const u64 *)va, gidx, gout, N, nlines);
    CK(hipEventRecord(ea));
    for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(gather8, dim3(N / 256), dim3(256), 0, 0, (const u64 *)va, gidx, gout, N, nlines);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb));
    g.gather = 3.0 * N / (ms * 1e-3) / 1e9;
    CK(hipEventRecord(ea));
    hipLaunchKernelGGL(block_streams, dim3(2048), dim3(256), 0, 0, (v4u *)va, bytes / 16 / 2048, sink);
    CK(hipEventRecord(eb)); CK(hipEventSynchronize(eb)); CK(hipEventElapsedTime(&ms, ea, eb));
    g.streams = 2.0 * bytes / (ms * 1e-3) / 1e9;
    return g;
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 4, maxn = argc > 2 ? atoi(argv[2]) : 66;
    const size_t gran = gib << 30;
    const bool plain = argc > 4 && argv[4][0] == 'm';
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMalloc(&sink, 4)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    CK(hipMalloc(&gidx, 8ull << 26)); CK(hipMalloc(&gout, 8ull << 26));
    hipLaunchKernelGGL(fill_idx, dim3((1u << 26) / 256), dim3(256), 0, 0, gidx, 1ull << 26);
    std::vector<hipMemGenericAllocationHandle_t> hs;
    std::vector<void *> vas;
    std::vector<double> rate;
    printf("granules of %zu GiB, %s.  Per granule: random 64-byte reads G/s non-temporal / ordinary / gather8 ; block streams GB/s\n", gib, plain ? "hipMalloc" : "hipMemCreate + hipMemMap");
    for (size_t k = 0; k < maxn; k++) {
        hipMemGenericAllocationHandle_t h = {};
        void *va = nullptr;
        if (plain) {
            if (hipMalloc(&va, gran) != hipSuccess) { (void)hipGetLastError(); break; }
        } else {
            if (hipMemCreate(&h, gran, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemAddressReserve(&va, gran, 0, nullptr, 0));
            CK(hipMemMap(va, gran, 0, h, 0));
            CK(hipMemSetAccess(va, gran, &acc, 1));
        }
        hs.push_back(h); vas.push_back(va);
        const Grade g = grade(va, gran);
        rate.push_back(g.gather);
        printf("%zu:%.1f/%.1f/%.1f;%.0f  ", k, g.nt, g.plain, g.gather, g.streams);
        if (k % 6 == 5) printf("\n");
        fflush(stdout);
    }
    printf("\n%zu granules\n", hs.size());
    if (plain) return 0;
    std::vector<size_t> order(hs.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return rate[x] > rate[y]; });
    for (size_t k = 0; k < hs.size(); k++) CK(hipMemUnmap(vas[k], gran));
    const size_t want = (argc > 3 ? (size_t)atoi(argv[3]) : 16) << 30, m = want / gran;
    if (m && m * 2 <= hs.size()) {
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, want, 0, nullptr, 0));
        for (int pass = 0; pass < 4; pass++) {
            for (size_t k = 0; k < m; k++) CK(hipMemMap((char *)va + k * gran, gran, 0, hs[(pass & 1) ? order[order.size() - 1 - k] : order[k]], 0));
            CK(hipMemSetAccess(va, want, &acc, 1));
            const Grade g = grade(va, want);
            printf("%zu GiB composed of the %s granules: non-temporal %.1f  ordinary %.1f  gather8 %.1f G/s ; block streams %.0f GB/s\n", want >> 30,
                   (pass & 1) ? "SLOWEST" : "FASTEST", g.nt, g.plain, g.gather, g.streams);
            CK(hipMemUnmap(va, want));
        }
        CK(hipMemAddressFree(va, want));
    }
    for (size_t k = 0; k < hs.size(); k++) { CK(hipMemRelease(hs[k])); CK(hipMemAddressFree(vas[k], gran)); }
    return 0;
}
