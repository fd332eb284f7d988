// placement.hip -- where the engine's big device buffers lie: allocation statistics, the parking lot for scratch pieces that were
// drawn and not used, and the placement of the chain scratch by GRADE.
#include "bsgs_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

static std::atomic<uint64_t> g_alloc_contiguous{0}, g_alloc_plain{0};
// bytes of big buffers this process obtained as physically contiguous memory / as ordinary pages (cumulative)
extern "C" int bsgs_alloc_stats(uint64_t *contiguous_bytes, uint64_t *plain_bytes)
{
    if (contiguous_bytes) *contiguous_bytes = g_alloc_contiguous.load();
    if (plain_bytes) *plain_bytes = g_alloc_plain.load();
    return BSGS_OK;
}
// Rejected scratch pieces are not freed while memory is plentiful: freeing tens of GiB makes the driver wipe them, which slows the GPU
// down in bursts for seconds (profiles/r02g_settling_after_tuning.log).  They wait here -- per process, any device -- until an engine is
// closed or an allocation fails (then everything parked on that device is released and the allocation is retried).
static std::mutex g_park_mu;
struct Parked { int device; void *p; uint64_t bytes; };
static std::vector<Parked> g_parked;
void park_release(int device)
{
    std::vector<void *> mine;
    {
        std::lock_guard<std::mutex> lk(g_park_mu);
        for (size_t k = 0; k < g_parked.size();) {
            if (g_parked[k].device == device) { mine.push_back(g_parked[k].p); g_parked[k] = g_parked.back(); g_parked.pop_back(); }
            else k++;
        }
    }
    for (void *p : mine) (void)bsgs_big_free(p);
}
uint64_t parked_bytes(int device)
{
    std::lock_guard<std::mutex> lk(g_park_mu);
    uint64_t n = 0;
    for (const Parked &x : g_parked) if (x.device == device) n += x.bytes;
    return n;
}
// What an allocation on the current device can really get: the driver's free figure plus what this process has parked there (parked pieces
// are handed back the moment an allocation needs them: malloc_or_unpark).  Every "does it fit" decision of the engine uses this, not the raw
// hipMemGetInfo, so that a second table upload or a second engine on the GPU is not talked into a slower layout by memory that is ours.
hipError_t bsgs_mem_available(size_t *avail, size_t *total)
{
    size_t fr = 0, tot = 0;
    const hipError_t e = hipMemGetInfo(&fr, &tot);
    if (e != hipSuccess) return e;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) fr += parked_bytes(dev);
    if (avail) *avail = fr;
    if (total) *total = tot;
    return hipSuccess;
}
static void park(int device, void *p, uint64_t bytes)
{
    std::lock_guard<std::mutex> lk(g_park_mu);
    g_parked.push_back({device, p, bytes});
}
static hipError_t malloc_or_unpark(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return e;
    bool any;
    { std::lock_guard<std::mutex> lk(g_park_mu); any = false; for (auto &x : g_parked) any |= x.device == dev; }
    if (!any) return e;
    park_release(dev);
    for (int k = 0; k < 24; k++) {                             // released memory is wiped before it can be allocated again
        e = hipMalloc(p, bytes);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::milliseconds(250));
    }
    return e;
}
// ---- buffers composed of physical chunks (hipMemCreate + hipMemMap) ------------------------------------------------------------------------
// Used for ONE thing: the bucket lines of a table above 40 GiB (bsgs_lines_malloc), where the placement needs every free 4 GiB chunk graded and
// the chosen ones in one contiguous range.  With hipMalloc that meant: allocate all pieces, free most of them, wait until the driver has wiped
// the ~200 GiB handed back (3.8 s at -w 34: profiles/r06e_*), allocate the range.  Chunks taken as physical handles are graded through a
// temporary mapping and then simply mapped where they are wanted: nothing is freed, nothing is wiped.  Every pointer handed out here is
// registered, so that one release call (bsgs_big_free) serves both kinds of memory.
struct VmBuf { size_t bytes; size_t chunk; std::vector<hipMemGenericAllocationHandle_t> handles; };
// Addresses.  Two facts of this stack, both measured (tools/experiments/vm_hint.hip, vm_release.hip; profiles/r06m_*):
//  * the memory of a released chunk comes back only when its ADDRESS RESERVATION is freed (hipMemUnmap + hipMemRelease inside a reservation that
//    lives on return nothing), so every mapping has a reservation of its own, freed with it;
//  * a freed reservation is handed out again at the very same address by the next hipMemAddressReserve, and a NEW mapping at a just-unmapped
//    address faults (the second engine of a two-engine host died mapping its chunks moments after the first had released some of its own).
// hipMemAddressReserve honours an address hint, so reservations are asked for at addresses that only ever go UP, in a part of the address space
// the runtime does not use on its own: no address is mapped twice in the life of the process.
static std::mutex g_vm_addr_mu;
static uintptr_t g_vm_cursor = 0x200000000000ull;                    // 32 TiB; the runtime's own reservations sit above 0x7000...
static void *vm_fresh_reserve(size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_vm_addr_mu);
    const size_t span = (bytes + (4ull << 30) - 1) & ~((4ull << 30) - 1);
    for (int tries = 0; tries < 16; tries++) {
        void *p = nullptr;
        const uintptr_t hint = g_vm_cursor;
        g_vm_cursor += span;
        if (hipMemAddressReserve(&p, bytes, 0, (void *)hint, 0) != hipSuccess) { (void)hipGetLastError(); g_vm_cursor += 1ull << 40; continue; }
        if ((uintptr_t)p == hint) return p;
        (void)hipMemAddressFree(p, bytes);                           // somebody lives there: the runtime chose an address of its own, which may be a recycled one
        g_vm_cursor += 1ull << 40;
    }
    return nullptr;
}
static std::mutex g_vm_mu;
static std::map<void *, VmBuf> g_vm;
static void vm_unmap_release(void *va, const VmBuf &b)
{
    for (size_t k = 0; k < b.handles.size(); k++) {
        (void)hipMemUnmap((char *)va + k * b.chunk, b.chunk);
        (void)hipMemRelease(b.handles[k]);
    }
    (void)hipMemAddressFree(va, b.bytes);
}
hipError_t bsgs_big_free(void *p)
{
    if (!p) return hipSuccess;
    VmBuf b;
    bool vm = false;
    {
        std::lock_guard<std::mutex> lk(g_vm_mu);
        auto it = g_vm.find(p);
        if (it != g_vm.end()) { b = it->second; g_vm.erase(it); vm = true; }
    }
    if (!vm) return hipFree(p);
    (void)hipDeviceSynchronize();
    vm_unmap_release(p, b);
    return hipSuccess;
}
static void vm_register(void *va, size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> handles)
{
    std::lock_guard<std::mutex> lk(g_vm_mu);
    g_vm[va] = VmBuf{bytes, chunk, std::move(handles)};
}

hipError_t bsgs_big_malloc(void **p, size_t bytes)
{
    // BSGS_CONTIGUOUS=1 asks for physically contiguous VRAM first (large page-table fragments).  It was tried as an explanation
    // of the run-to-run levels of the tile kernel (34.6 / 36.3 / 37.4 / 39.4 G on one box with one binary) and does not remove
    // them -- 14 runs with, 14 without: the same levels, the same mean (profiles/r02e_contiguous_allocation.log) -- so it stays off.
    static const int mode = getenv("BSGS_CONTIGUOUS") ? atoi(getenv("BSGS_CONTIGUOUS")) : 0;
    if (mode == 1 && bytes >= (64u << 20)) {
        if (hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess) { g_alloc_contiguous += bytes; return hipSuccess; }
        (void)hipGetLastError();                                   // refused (fragmented / too large): ordinary pages
    }
    const hipError_t e = malloc_or_unpark(p, bytes);
    if (e == hipSuccess && bytes >= (64u << 20)) g_alloc_plain += bytes;
    return e;
}
// ---- placement by grade ---------------------------------------------------------------------------------------------------
// An MI355X's 288 GB fall into THREE memory groups of ~89 GiB (the three ranks of its 12-high HBM3E stacks, by all appearances:
// tools/experiments/hbm_groups.hip, profiles/r02i_hbm_three_memory_groups.log): a latency-sensitive gather -- one 8-byte load per thread
// next to a coalesced index stream and a coalesced output stream -- runs at 38-39 G rows/s when its random reads share a group with its
// streams and at 42-43 G when they do not; 4 GiB allocations are almost always purely in one group (a few straddle: 40.5-41.4); which
// addresses belong to which group differs per box and per process.  The tile kernel obeys the same rule: its random probes (bucket
// lines) and its scratch streams (chain) cost +2...2.5 ms per launch for every 4 GiB they share a group with, and hipMalloc hands out
// whatever comes -- hence the run-to-run "levels" of 159...186 ms.  So the engine grades what it allocates:
//   * tables up to 40 GiB lie wherever hipMalloc put them (one group, sometimes two); the chain scratch -- pieces of <= 4 GiB, tile t in
//     piece t >> k -- is graded AGAINST THEM: the gather's random reads go all over the installed bucket lines while its two streams
//     run through the candidate piece, which is the kernel's own conflict in 2 ms; the highest-graded pieces are kept, the others
//     handed back at the end (released earlier they would be handed out again);
//   * larger tables (-w 34: 128 GiB) cannot avoid two groups, so before the lines are allocated one group is RESERVED piece by piece
//     (graded relative to a pair of buffers of the engine's own: "group 0"), the lines land in the other two, and the chain scratch
//     is then taken from the reserve.
// BSGS_CHAIN_PIECES=0 / BSGS_GRADED_LINES=0 switch back to plain allocations.
static __global__ void grade_gather_kernel(const unsigned long long *base, const unsigned long long *idx, unsigned long long *out, unsigned long long n, unsigned long long rows)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) out[i] = base[(idx[i] % rows) * 8];
}
static __global__ void grade_fill_kernel(unsigned long long *idx, unsigned long long n)
{
    const unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    if (i < n) { unsigned long long s = (i + 1) * 0x9E3779B97F4A7C15ull; s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32; idx[i] = s >> 8; }
}
// A grade is relative to where the gather's streams run: the same piece grades 39 with one pair of index/output buffers and 42 with another
// pair allocated elsewhere (profiles/r02i_grade_is_relative_to_the_graders_buffers.log).  grade_against() needs no buffers (the streams
// run through the candidate piece itself, the random reads through the installed table); grade() -- used only to reserve a group
// before a large table exists -- runs the streams through one 256 MiB buffer of the engine's own, kept for its life ("group 0").
struct PieceGrader {
    static constexpr unsigned long long N = 1ull << 24;
    bsgs_dev *d;
    explicit PieceGrader(bsgs_dev *dev) : d(dev) {}
    bool init()
    {
        if (d->grade_idx) return true;
        if (hipMalloc(&d->grade_idx, 2 * N * 8) != hipSuccess || hipEventCreate(&d->grade_ea) != hipSuccess || hipEventCreate(&d->grade_eb) != hipSuccess) {
            (void)hipGetLastError();
            release(d);
            return false;
        }
        d->grade_out = d->grade_idx + N;                         // both streams in one allocation: one memory group
        hipLaunchKernelGGL(grade_fill_kernel, dim3(N / 256), dim3(256), 0, d->stream, d->grade_idx, N);
        return hipGetLastError() == hipSuccess;
    }
    static void release(bsgs_dev *d)
    {
        if (d->grade_idx) (void)hipFree(d->grade_idx);
        if (d->grade_ea) (void)hipEventDestroy(d->grade_ea);
        if (d->grade_eb) (void)hipEventDestroy(d->grade_eb);
        d->grade_idx = d->grade_out = nullptr; d->grade_ea = d->grade_eb = nullptr;
    }
    // The kernel's own conflict, measured directly: random 64-byte-row reads all over `table` (the installed bucket lines) while the
    // index and output streams run through `piece` (free scratch: its first 256 MiB are overwritten).  Needs no buffers of its own.
    float grade_against(const void *table, uint64_t table_bytes, void *piece)
    {
        if (!d->grade_ea && (hipEventCreate(&d->grade_ea) != hipSuccess || hipEventCreate(&d->grade_eb) != hipSuccess)) { (void)hipGetLastError(); return 0.f; }
        unsigned long long *idx = (unsigned long long *)piece, *out = idx + N;
        float ms = 0.f;
        hipLaunchKernelGGL(grade_fill_kernel, dim3(N / 256), dim3(256), 0, d->stream, idx, N);
        hipLaunchKernelGGL(grade_gather_kernel, dim3(N / 256), dim3(256), 0, d->stream, (const unsigned long long *)table, idx, out, N, table_bytes / 64);
        if (hipEventRecord(d->grade_ea, d->stream) != hipSuccess) return 0.f;
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(grade_gather_kernel, dim3(N / 256), dim3(256), 0, d->stream, (const unsigned long long *)table, idx, out, N, table_bytes / 64);
        if (hipEventRecord(d->grade_eb, d->stream) != hipSuccess || hipEventSynchronize(d->grade_eb) != hipSuccess || hipEventElapsedTime(&ms, d->grade_ea, d->grade_eb) != hipSuccess || ms <= 0.f) return 0.f;
        return (float)(3.0 * N / (ms * 1e6));
    }
    float grade(const void *buf, uint64_t bytes)                 // G gathers/s over the first 4 GiB (or all) of buf; 0 on error
    {
        const unsigned long long rows = std::min<uint64_t>(bytes, 4ull << 30) / 64;
        float ms = 0.f;
        hipLaunchKernelGGL(grade_gather_kernel, dim3(N / 256), dim3(256), 0, d->stream, (const unsigned long long *)buf, d->grade_idx, d->grade_out, N, rows);
        if (hipEventRecord(d->grade_ea, d->stream) != hipSuccess) return 0.f;
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(grade_gather_kernel, dim3(N / 256), dim3(256), 0, d->stream, (const unsigned long long *)buf, d->grade_idx, d->grade_out, N, rows);
        if (hipEventRecord(d->grade_eb, d->stream) != hipSuccess || hipEventSynchronize(d->grade_eb) != hipSuccess || hipEventElapsedTime(&ms, d->grade_ea, d->grade_eb) != hipSuccess || ms <= 0.f) return 0.f;
        return (float)(3.0 * N / (ms * 1e6));
    }
};
void release_grader(bsgs_dev *d) { PieceGrader::release(d); }
void free_reserve(bsgs_dev *d)
{
    for (void *p : d->group0_reserve) (void)bsgs_big_free(p);
    d->group0_reserve.clear();
}
// hand back all but `keep` pieces of the reserve (the table builder found the GPU too full for its scratch: a fuller table than the reserve was sized for)
void trim_reserve(bsgs_dev *d, size_t keep)
{
    while (d->group0_reserve.size() > keep) { (void)bsgs_big_free(d->group0_reserve.back()); d->group0_reserve.pop_back(); }
}
void free_chain_pieces(bsgs_dev *d)
{
    for (u32x4 *p : d->chain_pieces) (void)bsgs_big_free(p);
    d->chain_pieces.clear();
    d->chain_piece_bytes = 0;
}
// ---- the grader's decision rule, free of any device (tests/test_abi.py drives it through bsgs_debug_grade_rule) ------------------------------
// Pieces are drawn one at a time and graded (G gathers/s against the installed bucket lines: a piece that shares a memory group with them grades
// 6-10 % lower, a straddler 2-3 %).  The rule says when to STOP drawing and WHICH pieces to keep; it makes no assumption about how many memory
// groups a GPU shows (three on the MI355X boxes seen so far; another partition mode may show one class or five):
//   stop    once `need` pieces are within 2 % of the best grade seen AND a separation was seen (some piece >= 5 % below the best: the best is then
//           a far class) -- or `need + 12` pieces were looked at without one (one class is all there is) -- or `need + extra_max` in any case;
//   keep    the `need` best (ties: the earlier allocation);
//   verdict separated = a piece >= 5 % below the best was seen.  Without separation the grades carry no information: the first `need` pieces in
//           allocation order are kept -- exactly what plain allocation would have given -- and the engine says so (chain_placement: separated 0).
struct GradeRule {
    static size_t good(const float *g, size_t n) { float best = 0.f; for (size_t k = 0; k < n; k++) best = std::max(best, g[k]); size_t c = 0; for (size_t k = 0; k < n; k++) c += g[k] >= 0.98f * best; return c; }
    static bool separated(const float *g, size_t n) { float best = 0.f; for (size_t k = 0; k < n; k++) best = std::max(best, g[k]); for (size_t k = 0; k < n; k++) if (g[k] <= 0.95f * best) return true; return false; }
    static bool stop(const float *g, size_t n, size_t need, size_t extra_max)
    {
        if (n < need) return false;
        if (n >= need + extra_max) return true;
        return good(g, n) >= need && (separated(g, n) || n >= need + 12);
    }
    // indices of the pieces to keep, best first
    static std::vector<size_t> keep(const float *g, size_t n, size_t need)
    {
        std::vector<size_t> idx(n);
        for (size_t k = 0; k < n; k++) idx[k] = k;
        if (separated(g, n)) std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return g[a] > g[b]; });
        idx.resize(std::min(need, n));
        return idx;
    }
};
// grades[0..n) = the grades in the order the pieces would be drawn; returns how many the rule draws, which it keeps, and whether it saw a separation
extern "C" int bsgs_debug_grade_rule(const float *grades, uint32_t n, uint32_t need, uint32_t extra_max, uint32_t *drawn, uint32_t *kept, uint32_t *separated)
{
    if (!grades || !drawn || !kept || !separated || !need) return bsgs_fail(BSGS_ERR_ARG, "null");
    uint32_t k = 0;
    while (k < n && !GradeRule::stop(grades, k, need, extra_max)) k++;
    *drawn = k;
    const std::vector<size_t> idx = GradeRule::keep(grades, k, need);
    for (size_t i = 0; i < idx.size(); i++) kept[i] = (uint32_t)idx[i];
    for (size_t i = idx.size(); i < need; i++) kept[i] = 0xFFFFFFFFu;
    *separated = GradeRule::separated(grades, k) ? 1u : 0u;
    return BSGS_OK;
}

// npieces buffers of piece_bytes each, preferring the gather-fast class; false = not enough memory (nothing is left allocated)
bool alloc_graded_pieces(bsgs_dev *d, size_t npieces, uint64_t piece_bytes)
{
    if (!d->group0_reserve.empty() && piece_bytes <= d->group0_piece_bytes && d->group0_reserve.size() >= npieces) {
        // a big table was installed with group 0 held back for exactly this (bsgs_lines_malloc): take the scratch from the reserve
        d->chain_pieces.clear();
        for (size_t k = 0; k < npieces; k++) { d->chain_pieces.push_back((u32x4 *)d->group0_reserve.back()); d->group0_reserve.pop_back(); }
        std::sort(d->chain_pieces.begin(), d->chain_pieces.end());
        // what the scratch does not need is PARKED (counted as available, released the moment an allocation needs it): handing it back costs a wipe -- 0.2 s
        // per 4 GiB chunk, synchronously, for chunk-mapped memory -- that a start-up should not pay
        for (void *p : d->group0_reserve) park(d->id, p, d->group0_piece_bytes);
        d->group0_reserve.clear();
        d->chain_graded = d->group0_graded; d->chain_rejected = d->group0_graded - (uint32_t)npieces;
        d->chain_grade_best = d->group0_grade_lo; d->chain_grade_worst = d->group0_grade_hi;
        d->chain_from_reserve = 1; d->chain_separated = 1; d->chain_grades.clear();
        return true;
    }
    free_reserve(d);
    d->chain_from_reserve = 0;
    struct Cand { void *p; float g; };
    std::vector<Cand> cands;
    PieceGrader G(d);
    const bool direct = d->lines && d->lines_bytes >= (1ull << 30) && piece_bytes >= (512ull << 20);     // grade against the installed bucket lines
    const bool can_grade = direct || G.init();
    const size_t extra = can_grade ? 24 + (getenv("BSGS_GRADE_MORE") ? 24 : 0) : 0;                  // at most this many more than needed (the slow class holds 16...22 granules of 4 GiB)
    // "enough pieces within 2 % of the best seen" only means something once a separation was seen: consecutive allocations tend to come from one
    // memory group, and with the 6 pieces a launch needs since round 3 (8 bytes per giant) the first six were sometimes all in the bucket lines'
    // group -- uniformly bad, all "within 2 % of the best", 175 ms per launch instead of 160 (profiles/r04f_*): GradeRule above.
    std::vector<float> grades;
    while (cands.size() < npieces + extra) {
        static const size_t grade_more = getenv("BSGS_GRADE_MORE") ? (size_t)atoi(getenv("BSGS_GRADE_MORE")) : 0;     // diagnostics: look at this many extra pieces
        if (cands.size() >= npieces + grade_more && (!can_grade || GradeRule::stop(grades.data(), grades.size(), npieces, extra))) break;
        size_t fr = 0, tot = 0;
        if (cands.size() >= npieces && (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < piece_bytes + (6ull << 30))) break;   // leave room for the rest of the engine
        void *p = nullptr;
        if ((cands.size() < npieces ? malloc_or_unpark(&p, piece_bytes) : hipMalloc(&p, piece_bytes)) != hipSuccess) { (void)hipGetLastError(); break; }
        const float g = !can_grade ? 1.f : direct ? G.grade_against(d->lines, d->lines_bytes, p) : G.grade(p, piece_bytes);
        cands.push_back({p, g});
        grades.push_back(g);
    }
    if (cands.size() < npieces) { for (const Cand &c : cands) (void)bsgs_big_free(c.p); return false; }
    // keep by the rule: the best `npieces` when a separation was seen, the first `npieces` drawn (= plain allocation) when the grades say nothing
    const bool sep = can_grade && GradeRule::separated(grades.data(), grades.size());
    {
        const std::vector<size_t> kept = GradeRule::keep(grades.data(), grades.size(), npieces);
        std::vector<Cand> ordered;
        std::vector<bool> taken(cands.size(), false);
        for (size_t k : kept) { ordered.push_back(cands[k]); taken[k] = true; }
        for (size_t k = 0; k < cands.size(); k++) if (!taken[k]) ordered.push_back(cands[k]);
        cands.swap(ordered);
    }
    d->chain_separated = sep ? 1 : 0;
    d->chain_grades.clear();
    for (const Cand &c : cands) d->chain_grades.push_back(c.g);          // kept pieces first
    if (can_grade && !sep && cands.size() > npieces)
        fprintf(stderr, "bsgs: chain scratch: %zu pieces graded %.1f ... %.1f G gathers/s, no separation of 5 %% between them -- this GPU shows one memory class here; "
                        "the scratch is placed as plain allocation would (expect the launch time to vary with placement)\n", cands.size(),
                *std::min_element(grades.begin(), grades.end()), *std::max_element(grades.begin(), grades.end()));
    (void)hipStreamSynchronize(d->stream);
    d->chain_pieces.clear();
    for (size_t k = 0; k < npieces; k++) d->chain_pieces.push_back((u32x4 *)cands[k].p);
    std::sort(d->chain_pieces.begin(), d->chain_pieces.end());
    {
        size_t fr = 0, tot = 0;
        const bool plenty = hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= (96ull << 30);
        for (size_t k = npieces; k < cands.size(); k++) { if (plenty) park(d->id, cands[k].p, piece_bytes); else (void)bsgs_big_free(cands[k].p); }
    }
    d->chain_graded = (uint32_t)cands.size(); d->chain_rejected = (uint32_t)(cands.size() - npieces);
    d->chain_grade_best = d->chain_grade_worst = cands[0].g;
    for (size_t k = 0; k < npieces; k++) { d->chain_grade_best = std::max(d->chain_grade_best, cands[k].g); d->chain_grade_worst = std::min(d->chain_grade_worst, cands[k].g); }
    if (getenv("BSGS_TUNE_VERBOSE")) {
        fprintf(stderr, "[chain pieces] %zu x %.2f GiB, graded %zu:", npieces, piece_bytes / 1073741824.0, cands.size());
        for (size_t k = 0; k < cands.size(); k++) fprintf(stderr, "%s%.1f", k == npieces ? " | rejected " : " ", cands[k].g);
        fprintf(stderr, "\n");
    }
    return true;
}

// Large table, by physical chunks: free 4 GiB chunks are taken as handles one at a time (chunks this process parked earlier first), mapped at an address
// of their own and graded against the grader's buffers ("group 0" = the memory group THEY lie in), until enough were seen: as many outside group 0 as the
// lines need, eight inside it.  The lines are then composed of the chunks furthest from group 0, up to 24 group-0 chunks stay mapped as they are -- the
// pieces the chain scratch will be taken from (alloc_graded_pieces) -- and what is left over is parked (released the moment an allocation needs it).
static hipError_t lines_malloc_chunks(bsgs_dev *d, PieceGrader &G, void **out, size_t bytes)
{
    const size_t chunk = 4ull << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = d->id;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran || chunk % gran) return hipErrorNotSupported;
    // Access: the owning GPU -- and every GPU that can reach it over xGMI.  A mapping made with the virtual-memory API does NOT inherit hipDeviceEnablePeerAccess:
    // without the peers here, the replicas of a -w 34 table (bsgs_broadcast_tables, RCCL inside one process, the receive buffers of bsgs_alloc_table_ext_recv)
    // would fault on the first cross-device copy into or out of these lines.  If the driver refuses the peers the owner alone is granted and the
    // table stays usable on its own GPU (BSGS_CHUNK_PEER_ACCESS=0 does the same on purpose).
    std::vector<hipMemAccessDesc> accs(1);
    accs[0].location = prop.location;
    accs[0].flags = hipMemAccessFlagsProtReadWrite;
    {
        static const bool peers_on = !(getenv("BSGS_CHUNK_PEER_ACCESS") && atoi(getenv("BSGS_CHUNK_PEER_ACCESS")) == 0);
        int ndev = 0;
        if (peers_on && hipGetDeviceCount(&ndev) == hipSuccess)
            for (int p = 0; p < ndev; p++) {
                int can = 0;
                if (p == d->id || hipDeviceCanAccessPeer(&can, p, d->id) != hipSuccess || !can) continue;
                hipMemAccessDesc a = {};
                a.location.type = hipMemLocationTypeDevice; a.location.id = p; a.flags = hipMemAccessFlagsProtReadWrite;
                accs.push_back(a);
            }
        (void)hipGetLastError();
    }
    auto set_access = [&](void *va, size_t bytes) {
        if (accs.size() > 1 && hipMemSetAccess(va, bytes, accs.data(), accs.size()) == hipSuccess) return hipSuccess;
        if (accs.size() > 1) { (void)hipGetLastError(); fprintf(stderr, "bsgs: peer access to mapped bucket lines refused by the driver: this table cannot be replicated device-to-device\n"); accs.resize(1); }
        return hipMemSetAccess(va, bytes, accs.data(), 1);
    };
    struct C { hipMemGenericAllocationHandle_t h; void *va; float g; };
    std::vector<C> all;
    const size_t need = (bytes + chunk - 1) / chunk;
    const bool verbose = getenv("BSGS_BUILD_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    auto drop = [&](C &c) { if (c.va) { (void)hipMemUnmap(c.va, chunk); (void)hipMemAddressFree(c.va, chunk); } (void)hipMemRelease(c.h); c.va = nullptr; };
    for (void *p : d->group0_reserve) park(d->id, p, d->group0_piece_bytes);
    d->group0_reserve.clear();
    // chunks this process parked earlier (an earlier table, the other engine on this GPU) are candidates like any other: already taken, already mapped
    std::vector<C> adopted;
    {
        std::vector<Parked> mine;
        {
            std::lock_guard<std::mutex> lk(g_park_mu);
            for (size_t k = 0; k < g_parked.size();) {
                if (g_parked[k].device == d->id) { mine.push_back(g_parked[k]); g_parked[k] = g_parked.back(); g_parked.pop_back(); }
                else k++;
            }
        }
        for (const Parked &x : mine) {
            bool vm = false;
            C c{};
            if (x.bytes == chunk) {
                std::lock_guard<std::mutex> lk(g_vm_mu);
                auto it = g_vm.find(x.p);
                if (it != g_vm.end() && it->second.handles.size() == 1 && it->second.bytes == chunk) { c.h = it->second.handles[0]; c.va = x.p; g_vm.erase(it); vm = true; }
            }
            if (vm) adopted.push_back(c);
            else (void)bsgs_big_free(x.p);
        }
    }
    float top = 0.f;
    for (;;) {
        size_t fr = 0, tot = 0;
        C c{};
        if (!adopted.empty()) {
            c = adopted.back(); adopted.pop_back();
            c.g = G.grade(c.va, chunk);
            top = std::max(top, c.g);
            all.push_back(c);
            size_t far = 0, near = 0;
            for (const C &x : all) { if (x.g <= 0.93f * top) near++; else far++; }
            if (far >= need && (near >= 8 || all.size() >= need + 24)) break;
            continue;
        }
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < chunk + (2ull << 30)) break;
        if (hipMemCreate(&c.h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        if (!(c.va = vm_fresh_reserve(chunk))) { (void)hipMemRelease(c.h); break; }
        if (hipMemMap(c.va, chunk, 0, c.h, 0) != hipSuccess || hipMemSetAccess(c.va, chunk, accs.data(), 1) != hipSuccess) {      // a temporary mapping, for grading: the owner only
            (void)hipGetLastError();
            (void)hipMemAddressFree(c.va, chunk); (void)hipMemRelease(c.h);
            break;
        }
        c.g = G.grade(c.va, chunk);
        top = std::max(top, c.g);
        all.push_back(c);
        // enough: `need` chunks outside group 0 for the lines and eight inside it for the chain scratch (24 GiB at the headline geometry = six).  Taking
        // a chunk costs up to 50 ms (the driver clears memory it hands out): the rest of the free memory is left alone
        size_t far = 0, near = 0;
        for (const C &x : all) { if (x.g <= 0.93f * top) near++; else far++; }
        if (far >= need && (near >= 8 || all.size() >= need + 24)) break;        // (a second engine on the GPU finds most of group 0 taken: it does not go on for ever)
    }
    (void)hipStreamSynchronize(d->stream);
    for (C &c : adopted) { vm_register(c.va, chunk, chunk, {c.h}); park(d->id, c.va, chunk); }      // not looked at: parked again
    adopted.clear();
    if (all.size() < need) { for (C &c : all) drop(c); return hipErrorOutOfMemory; }
    // lines: the `need` chunks furthest from group 0 (highest grades); group 0 = grade <= 0.93 x top
    std::vector<size_t> order(all.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return all[a].g > all[b].g; });
    std::vector<bool> used(all.size(), false);
    void *big = vm_fresh_reserve(need * chunk);
    if (!big) { for (C &c : all) drop(c); return hipErrorOutOfMemory; }
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t in_group0 = 0;
    bool ok = true;
    for (size_t k = 0; k < need && ok; k++) {
        C &c = all[order[k]];
        used[order[k]] = true;
        in_group0 += c.g <= 0.93f * top;
        // a chunk moves: its temporary mapping goes (address and all: a second mapping at a just-unmapped address faults), then it is mapped into the range
        ok = hipMemUnmap(c.va, chunk) == hipSuccess && hipMemAddressFree(c.va, chunk) == hipSuccess;
        c.va = nullptr;
        ok = ok && hipMemMap((char *)big + k * chunk, chunk, 0, c.h, 0) == hipSuccess;
        if (ok) handles.push_back(c.h);
    }
    ok = ok && set_access(big, need * chunk) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        for (size_t k = 0; k < handles.size(); k++) (void)hipMemUnmap((char *)big + k * chunk, chunk);
        (void)hipMemAddressFree(big, need * chunk);
        for (size_t k = 0; k < all.size(); k++) { if (used[k]) { all[k].va = nullptr; (void)hipMemRelease(all[k].h); } else drop(all[k]); }
        return hipErrorUnknown;
    }
    vm_register(big, need * chunk, chunk, handles);
    // the reserve for the chain scratch: group-0 chunks that the lines did not need, as mapped pieces; everything else goes back
    float lo = 1e30f, hi = 0.f;
    size_t released = 0;
    // how much is held back for the chain scratch: 24 chunks (96 GiB) while memory is plentiful -- the scratch is picked from them by grade --, 8 chunks (the 24 GiB of a full
    // launch + two) when the table took most of the GPU (-w 35: 48 of 66 chunks): everything held back is invisible to the table builder, which runs next and needs 8 GiB itself
    const size_t reserve_cap = all.size() - need >= 30 ? 24 : 8;
    for (size_t k = 0; k < all.size(); k++) {
        if (used[k]) continue;
        C &c = all[k];
        if (c.g <= 0.93f * top && d->group0_reserve.size() < reserve_cap) {
            vm_register(c.va, chunk, chunk, {c.h});
            d->group0_reserve.push_back(c.va);
            lo = std::min(lo, c.g); hi = std::max(hi, c.g);
        } else {
            // not needed: parked, not released -- memory that is handed back is wiped before anyone can have it again, and an allocation right after this one
            // (the overflow set, the chain scratch of a second engine) would find the GPU "full"; parked chunks are released the moment an allocation fails
            vm_register(c.va, chunk, chunk, {c.h});
            park(d->id, c.va, chunk);
            released++;
        }
    }
    d->group0_piece_bytes = chunk; d->group0_graded = (uint32_t)all.size(); d->group0_grade_lo = lo > 1e29f ? 0.f : lo; d->group0_grade_hi = hi;
    if (verbose || getenv("BSGS_TUNE_VERBOSE"))
        fprintf(stderr, "[place] %.0f GiB of lines composed of %zu chunks of 4 GiB (%zu of them in the grader's group), %zu chunks graded (top %.1f) in %.0f ms, "
                        "%zu of group 0 held back for the chain scratch (%.1f...%.1f), %zu parked\n", bytes / 1073741824.0, need, in_group0, all.size(), top, ms_since(t0),
                d->group0_reserve.size(), d->group0_grade_lo, hi, released);
    *out = big;
    g_alloc_plain += bytes;
    return hipSuccess;
}

// The bucket lines (see "placement by grade" above): plain up to 40 GiB; larger tables get one memory group reserved for the chain scratch first.
hipError_t bsgs_lines_malloc(bsgs_dev *d, void **out, size_t bytes)
{
    static const bool on = !(getenv("BSGS_GRADED_LINES") && atoi(getenv("BSGS_GRADED_LINES")) == 0);
    if (!on || !d || bytes <= (40ull << 30)) return bsgs_big_malloc(out, bytes);      // anywhere: the scratch pieces are graded against these very lines
    {
        // Lines that cover more than two of the three memory groups (-w 35: 192 GiB of 268) leave no group to reserve: holding pieces back then only pushes the
        // overflow set and the scratch into the same corner.  Measured on one box (profiles/r07n_w35_reserved_group_or_not.log): 3 * 2^30 lines of 64 bytes 28.6 G
        // with a reserved group and 36.7-37.1 G without, 1.5 * 2^30 lines of 128 bytes 34.2-35.2 G against 35.7-35.8 G; and the placement is 1-4 s shorter.
        size_t fr = 0, tot = 0;
        static const bool anyway = getenv("BSGS_RESERVE_ANYWAY") && atoi(getenv("BSGS_RESERVE_ANYWAY")) != 0;      // A-B only: the reserve also for lines above 0.6 of the HBM
        if (!anyway && hipMemGetInfo(&fr, &tot) == hipSuccess && (double)bytes > 0.6 * (double)tot) {
            if (getenv("BSGS_BUILD_VERBOSE")) fprintf(stderr, "[place] %.0f GiB of lines cover more than two memory groups of this GPU: no group reserved, one plain allocation\n", bytes / 1073741824.0);
            free_reserve(d);
            return bsgs_big_malloc(out, bytes);
        }
    }
    PieceGrader G(d);
    if (!G.init()) return bsgs_big_malloc(out, bytes);
    static const bool chunks_on = !(getenv("BSGS_CHUNK_LINES") && atoi(getenv("BSGS_CHUNK_LINES")) == 0);
    if (chunks_on) {
        if (lines_malloc_chunks(d, G, out, bytes) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();                                  // the virtual-memory calls are not there / failed: the hipMalloc walk below
    }
    {
        // Large table: walk through the free memory in 4 GiB pieces, keep every piece of group 0 (low grade) as the reserve the chain
        // scratch will be taken from, give the others back, THEN allocate the lines: they land in the other two groups.
        free_reserve(d);
        park_release(d->id);                                      // everything free is about to be graded: parked pieces belong to that walk
        const uint64_t piece = 4ull << 30;
        struct P { void *p; float g; };
        std::vector<P> all;
        float top = 0.f;
        const bool verbose = getenv("BSGS_BUILD_VERBOSE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
        auto t_walk = now();
        double ms_malloc = 0, ms_grade = 0;
        for (;;) {
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < piece + (2ull << 30)) break;
            void *p = nullptr;
            auto t0 = now();
            if (hipMalloc(&p, piece) != hipSuccess) { (void)hipGetLastError(); break; }
            ms_malloc += ms_since(t0); t0 = now();
            const float g = G.grade(p, piece);
            ms_grade += ms_since(t0);
            top = std::max(top, g);
            all.push_back({p, g});
        }
        (void)hipStreamSynchronize(d->stream);
        if (verbose) {
            fprintf(stderr, "[place] walk: %zu pieces of 4 GiB in %.0f ms (hipMalloc %.0f ms, grading %.0f ms); grades in allocation order:", all.size(), ms_since(t_walk), ms_malloc, ms_grade);
            for (const P &x : all) fprintf(stderr, " %.1f", x.g);
            fprintf(stderr, "\n");
        }
        auto t_free = now();
        float lo = 1e30f, hi = 0.f;
        for (const P &x : all) {
            if (x.g <= 0.93f * top && d->group0_reserve.size() < 24) { d->group0_reserve.push_back(x.p); lo = std::min(lo, x.g); hi = std::max(hi, x.g); }
            else (void)hipFree(x.p);
        }
        if (verbose) fprintf(stderr, "[place] %zu pieces held back, %zu handed back in %.0f ms\n", d->group0_reserve.size(), all.size() - d->group0_reserve.size(), ms_since(t_free));
        auto t_big = now();
        d->group0_piece_bytes = piece; d->group0_graded = (uint32_t)all.size(); d->group0_grade_lo = lo > 1e29f ? 0.f : lo; d->group0_grade_hi = hi;
        if (getenv("BSGS_TUNE_VERBOSE")) fprintf(stderr, "[lines] %.1f GiB: %zu pieces graded (top %.1f), %zu of group 0 held back for the chain scratch (%.1f...%.1f)\n",
                                                 bytes / 1073741824.0, all.size(), top, d->group0_reserve.size(), d->group0_grade_lo, hi);
        // the pieces just handed back are wiped by the driver before they can be allocated again: an allocation this large may have to wait
        auto patient = [&](int tries) {
            hipError_t e = hipErrorOutOfMemory;
            for (int k = 0; k < tries && e != hipSuccess; k++) {
                e = bsgs_big_malloc(out, bytes);
                if (e != hipSuccess) { (void)hipGetLastError(); std::this_thread::sleep_for(std::chrono::milliseconds(250)); }
            }
            return e;
        };
        hipError_t e = patient(32);
        if (e != hipSuccess) { free_reserve(d); e = patient(32); }         // not with the reserve in the way: without it
        if (verbose) fprintf(stderr, "[place] hipMalloc of the %.0f GiB of lines: %.0f ms\n", bytes / 1073741824.0, ms_since(t_big));
        return e;
    }
}

