// graded_alloc.hip -- EXPERIMENT, NOT PART OF THE LIBRARY (negative result, kept for the record; it was wired into bsgs_big_malloc for
// one A/B on the GPU and taken out again).
//
// Idea: on every MI355X box tried, 4 GiB granules of physical memory fall in two classes for a latency-sensitive gather (43.2 vs 38.8 G
// gathers/s; tools/experiments/hbm_map.hip, profiles/r02g_hbm_map_4GiB_and_1GiB_granules.log), and the tile kernel's launch time takes one
// of several levels depending on where its chain scratch and bucket lines were placed (profiles/r02e_*.log).  So: take the memory granule
// by granule (hipMemCreate), grade every granule with that gather, compose the buffers of the best granules (hipMemMap into one range).
//
// Result (profiles/r02g_graded_allocation_ab.log): buffers composed of the best-graded granules ran the tile kernel at 178...183 ms per
// 192-tile launch in four processes out of four -- the SLOW end of the 159...186 ms range -- plain hipMalloc in the same session at 172
// and 175 ms, buffers composed of the worst-graded granules at 176 and 185 ms.  The gather's classes are real but they are not what
// the tile kernel's levels follow, and hipMemMap-composed buffers are no better than hipMalloc'ed ones.  Two things learned on the way:
// mapping a second handle at an address that was just unmapped faults (stale translation): every candidate needs an address of its own;
// hipMemAddressReserve ignores its alignment argument (and the virtual alignment makes no difference: hbm_align.hip).
#include "bsgs_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr size_t GRANULE = 4ull << 30;            // the classes are clean at this size (1 GiB granules are not: hbm_map)
constexpr uint64_t GATHERS = 1ull << 24;          // per grading launch: 0.4 ms
constexpr size_t EXTRA_GRANULES = 28;             // how many more than needed may be tried (112 GiB: more than the slow class holds)

struct Graded {
    size_t bytes = 0;                              // mapped (whole granules, tail granule may be smaller)
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};
std::mutex g_mu;
std::map<void *, Graded> g_live;
uint64_t g_graded_bytes = 0, g_tried = 0, g_rejected = 0;
double g_best = 0, g_worst = 0;

// one 8-byte load per thread from a random 64-byte row of the granule, beside a coalesced index stream and a coalesced output stream
__global__ void grade_gather(const unsigned long long *base, const unsigned long long *idx, unsigned long long *out, unsigned long long n, unsigned long long rows)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = base[(idx[i] % rows) * 8];
}
__global__ void grade_fill(unsigned long long *idx, unsigned long long n)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) { unsigned long long s = (i + 1) * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32; idx[i] = s >> 8; }
}

struct Grader {
    unsigned long long *idx = nullptr, *out = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t init()
    {
        hipError_t e;
        if ((e = hipMalloc(&idx, GATHERS * 8)) != hipSuccess) return e;
        if ((e = hipMalloc(&out, GATHERS * 8)) != hipSuccess) return e;
        if ((e = hipEventCreate(&a)) != hipSuccess) return e;
        if ((e = hipEventCreate(&b)) != hipSuccess) return e;
        hipLaunchKernelGGL(grade_fill, dim3(GATHERS / 256), dim3(256), 0, 0, idx, GATHERS);
        return hipGetLastError();
    }
    ~Grader()
    {
        if (idx) (void)hipFree(idx);
        if (out) (void)hipFree(out);
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
    }
    // gathers per nanosecond over [va, va + bytes)
    hipError_t rate(void *va, size_t bytes, double *r)
    {
        hipError_t e;
        float ms = 0;
        hipLaunchKernelGGL(grade_gather, dim3(GATHERS / 256), dim3(256), 0, 0, (const unsigned long long *)va, idx, out, GATHERS, bytes / 64);
        if ((e = hipEventRecord(a, 0)) != hipSuccess) return e;
        for (int rep = 0; rep < 3; rep++)
            hipLaunchKernelGGL(grade_gather, dim3(GATHERS / 256), dim3(256), 0, 0, (const unsigned long long *)va, idx, out, GATHERS, bytes / 64);
        if ((e = hipEventRecord(b, 0)) != hipSuccess) return e;
        if ((e = hipEventSynchronize(b)) != hipSuccess) return e;
        if ((e = hipEventElapsedTime(&ms, a, b)) != hipSuccess) return e;
        *r = 3.0 * GATHERS / (ms * 1e6);
        return hipSuccess;
    }
};

int graded_mode()
{
    static const int mode = getenv("BSGS_GRADED") ? atoi(getenv("BSGS_GRADED")) : 1;
    return mode;
}

hipError_t graded_malloc(void **p, size_t bytes)
{
    int dev = 0;
    hipError_t e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t page = 0;
    if ((e = hipMemGetAllocationGranularity(&page, &prop, hipMemAllocationGranularityRecommended)) != hipSuccess) return e;
    if (page == 0 || (GRANULE % page) != 0) return hipErrorNotSupported;
    const size_t total = (bytes + page - 1) / page * page;
    const size_t full = total / GRANULE, tail = total - full * GRANULE;
    Grader g;
    if ((e = g.init()) != hipSuccess) return e;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    struct Cand { hipMemGenericAllocationHandle_t h; double rate; };
    std::vector<Cand> cands;
    // every candidate is graded at an address of its own (one reservation, a granule apart)
    const size_t slots = full + EXTRA_GRANULES;
    void *scratch = nullptr;
    if ((e = hipMemAddressReserve(&scratch, slots * GRANULE, 0, nullptr, 0)) != hipSuccess) return e;
    const bool want_slow = graded_mode() == 2;
    auto good = [&](double best) {                     // candidates within 4 % of the best seen (the classes are 10 % apart)
        size_t n = 0;
        for (const Cand &c : cands) n += c.rate >= 0.96 * best;
        return n;
    };
    double best = 0;
    hipError_t fatal = hipSuccess;
    while (cands.size() < slots) {
        if (!want_slow && good(best) >= full) break;
        Cand c = {};
        if (hipMemCreate(&c.h, GRANULE, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }     // out of memory: make do
        void *at = (char *)scratch + cands.size() * GRANULE;
        if ((e = hipMemMap(at, GRANULE, 0, c.h, 0)) != hipSuccess) { fatal = e; (void)hipMemRelease(c.h); break; }
        if ((e = hipMemSetAccess(at, GRANULE, &acc, 1)) != hipSuccess || (e = g.rate(at, GRANULE, &c.rate)) != hipSuccess) {
            fatal = e; (void)hipMemUnmap(at, GRANULE); (void)hipMemRelease(c.h); break;
        }
        best = std::max(best, c.rate);
        cands.push_back(c);
    }
    (void)hipDeviceSynchronize();
    for (size_t k = 0; k < cands.size(); k++) (void)hipMemUnmap((char *)scratch + k * GRANULE, GRANULE);
    (void)hipMemAddressFree(scratch, slots * GRANULE);
    auto drop_all = [&]() { for (const Cand &c : cands) (void)hipMemRelease(c.h); };
    if (fatal != hipSuccess) { drop_all(); return fatal; }
    if (cands.size() < full) { drop_all(); return hipErrorOutOfMemory; }
    std::sort(cands.begin(), cands.end(), [&](const Cand &x, const Cand &y) { return want_slow ? x.rate < y.rate : x.rate > y.rate; });
    Graded G;
    hipMemGenericAllocationHandle_t tail_h = {};
    if (tail) {
        // the tail (less than a granule) is taken ungraded AFTER the choice, while the rejected granules are still held
        if (hipMemCreate(&tail_h, tail, &prop, 0) != hipSuccess) { (void)hipGetLastError(); drop_all(); return hipErrorOutOfMemory; }
    }
    void *va = nullptr;
    if ((e = hipMemAddressReserve(&va, total, 0, nullptr, 0)) != hipSuccess) { drop_all(); if (tail) (void)hipMemRelease(tail_h); return e; }
    for (size_t k = 0; k < full && e == hipSuccess; k++) {
        e = hipMemMap((char *)va + k * GRANULE, GRANULE, 0, cands[k].h, 0);
        if (e == hipSuccess) { G.handles.push_back(cands[k].h); G.sizes.push_back(GRANULE); }
    }
    if (e == hipSuccess && tail) {
        e = hipMemMap((char *)va + full * GRANULE, tail, 0, tail_h, 0);
        if (e == hipSuccess) { G.handles.push_back(tail_h); G.sizes.push_back(tail); }
    }
    if (e == hipSuccess) e = hipMemSetAccess(va, total, &acc, 1);
    if (e != hipSuccess) {
        size_t off = 0;
        for (size_t k = 0; k < G.handles.size(); k++) { (void)hipMemUnmap((char *)va + off, G.sizes[k]); off += G.sizes[k]; }
        (void)hipMemAddressFree(va, total);
        drop_all();
        if (tail) (void)hipMemRelease(tail_h);
        return e;
    }
    for (size_t k = full; k < cands.size(); k++) (void)hipMemRelease(cands[k].h);       // the rejected ones go back to the driver now
    G.bytes = total;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_live[va] = G;
        g_graded_bytes += total;
        g_tried += cands.size();
        g_rejected += cands.size() - full;
        if (full) {
            g_best = std::max(g_best, cands[0].rate);
            const double w = cands[full - 1].rate;
            g_worst = g_worst == 0 ? w : std::min(g_worst, w);
        }
    }
    if (getenv("BSGS_GRADED_VERBOSE")) {
        fprintf(stderr, "[graded] %.1f GiB: %zu granules tried, kept %zu at %.1f ... %.1f G gathers/s, rejected:", total / 1073741824.0, cands.size(), full,
                full ? cands[0].rate : 0.0, full ? cands[full - 1].rate : 0.0);
        for (size_t k = full; k < cands.size(); k++) fprintf(stderr, " %.1f", cands[k].rate);
        fprintf(stderr, "\n");
    }
    *p = va;
    return hipSuccess;
}

}  // namespace

// [0] bytes composed of graded granules, [1] granules tried, [2] granules rejected, [3]/[4] best / worst kept grade (gathers per us)
extern "C" int bsgs_graded_stats(uint64_t out[5])
{
    if (!out) return -1;
    std::lock_guard<std::mutex> lk(g_mu);
    out[0] = g_graded_bytes; out[1] = g_tried; out[2] = g_rejected; out[3] = (uint64_t)(g_best * 1000); out[4] = (uint64_t)(g_worst * 1000);
    return 0;
}

hipError_t bsgs_graded_malloc(void **p, size_t bytes)
{
    if (graded_mode() == 0 || bytes < GRANULE) return hipErrorNotSupported;
    return graded_malloc(p, bytes);
}

// frees a pointer of bsgs_big_malloc (graded or plain)
hipError_t bsgs_big_free(void *p)
{
    if (!p) return hipSuccess;
    Graded G;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) return hipFree(p);
        G = it->second;
        g_live.erase(it);
    }
    (void)hipDeviceSynchronize();
    size_t off = 0;
    for (size_t k = 0; k < G.handles.size(); k++) {
        (void)hipMemUnmap((char *)p + off, G.sizes[k]);
        (void)hipMemRelease(G.handles[k]);
        off += G.sizes[k];
    }
    return hipMemAddressFree(p, G.bytes);
}
