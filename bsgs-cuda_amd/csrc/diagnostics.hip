// diagnostics.hip -- measurement and parity instruments of libbsgs_hip.so: placement tuning and its reports, the probe digest, phase timing, the per-XCD profile,
// the device arithmetic selftests and the roofline denominators.  Nothing here is on the search path.
#include "bsgs_internal.h"
#include "support_kernels.hip.h"
#include "host_secp.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// how the chain scratch of the default kernel is laid out: info[0] pieces (0 = one buffer), [1] tiles per piece, [2] pieces graded by the
// last allocation, [3] pieces handed back, [4] 1 = taken from the memory group reserved while a large table was installed;
// grade[0], grade[1] = grades of the pieces kept, best and worst (G gathers/s; against the installed bucket lines: higher = further from them)
extern "C" int bsgs_chain_placement(bsgs_dev *d, uint32_t info[5], float grade[2])
{
    if (!d || !info || !grade) return fail(BSGS_ERR_ARG, "null");
    info[0] = (uint32_t)d->chain_pieces.size(); info[1] = d->chain_pieces.empty() ? 0 : 1u << d->chain_piece_log;
    info[2] = d->chain_graded; info[3] = d->chain_rejected; info[4] = d->chain_from_reserve;
    grade[0] = d->chain_grade_best; grade[1] = d->chain_grade_worst;
    return BSGS_OK;
}

// every grade the last graded allocation of the chain scratch saw (G gathers/s), the kept pieces first; *separated = 1 when two classes were seen
extern "C" int bsgs_chain_grades(bsgs_dev *d, float *grades, uint32_t cap, uint32_t *n, uint32_t *separated)
{
    if (!d || !n) return fail(BSGS_ERR_ARG, "null");
    *n = (uint32_t)d->chain_grades.size();
    if (separated) *separated = d->chain_separated;
    for (uint32_t k = 0; grades && k < cap && k < *n; k++) grades[k] = d->chain_grades[k];
    return BSGS_OK;
}

// Start-up tuning of WHERE the chain scratch and the bucket lines lie.  The launch time of the tile kernel depends on the physical
// memory the driver happened to hand out for these two buffers (159 ... 186 ms for the same 192 tiles); every allocation
// re-draws it and the level then persists for the life of the allocation (profiles/r02g_tuned_placement_persists.log).  So: time
// launches of walk tiles on up to `candidates` allocations of the scratch -- all held at once, so that every one is different memory --
// keep the fastest, free the rest; then the same for the bucket lines (device-to-device copies).  Freeing tens of GiB slows the GPU
// down for a second or two (the driver wipes released memory), so the call ends by running launches until the chosen time is back.
// Needs the walk, the giants and the table; the tiles' hits are discarded; a buffer is left alone (not an error) when the free memory
// does not hold a second copy of it.  ms_out[0..candidates) = launch times on the scratch candidates, ms_out[candidates..2*candidates) on
// the line candidates (0 = not tried); chosen[0], chosen[1] = indices kept.
extern "C" int bsgs_tune_placement(bsgs_dev *d, uint32_t candidates, float *ms_out, uint32_t chosen[2], float *final_ms)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->walk_set) return fail(BSGS_ERR_STATE, "bsgs_set_walk first");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    if (candidates == 0 || candidates > 16) return fail(BSGS_ERR_ARG, "1..16 candidates");
    HIPCHK(hipSetDevice(d->id));
    const uint32_t tpl = bsgs_auto_tiles_per_launch(d);
    auto launch = [&](float *ms) -> int {
        int rc = bsgs_enqueue_walk(d, 0, tpl);
        if (rc) return rc;
        uint32_t n = 0;
        rc = bsgs_collect(d, nullptr, 0, &n, ms);
        return rc == BSGS_ERR_OVERFLOW ? BSGS_OK : rc;
    };
    auto timed = [&](float *ms) -> int {                   // one warm launch, then two timed ones
        float t[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; k++) { int rc = launch(&t[k]); if (rc) return rc; }
        *ms = (t[1] + t[2]) / 2;
        return BSGS_OK;
    };
    auto room_for = [&](uint64_t bytes) { size_t fr = 0, tot = 0; return bsgs_mem_available(&fr, &tot) == hipSuccess && fr >= bytes + (8ull << 30); };
    if (ms_out) for (uint32_t k = 0; k < 2 * candidates; k++) ms_out[k] = 0.f;
    float best_ms = 0.f;
    int rc = BSGS_OK;
    // the allocations before this call (graded bucket lines, graded scratch pieces) handed memory back too: wait until eight launches in a
    // row are within 1 % of the fastest seen, 8 s at most, before anything is compared
    {
        float lo = 1e30f, t = 0.f;
        for (int k = 0, calm = 0; k < 48 && calm < 8; k++) {
            if ((rc = launch(&t))) return rc;
            if (t < lo * 0.99f) { lo = t; calm = 0; }
            else if (t <= lo * 1.01f) { calm++; lo = std::min(lo, t); }
            else calm = 0;
        }
    }
    // ---- chain scratch
    {
        std::vector<u32x4 *> held;
        std::vector<float> ms;
        float t = 0.f;
        if ((rc = timed(&t))) return rc;                   // allocates the scratch if this is the first launch
        held.push_back(d->chain); ms.push_back(t);
        const uint64_t bytes = d->chain_bytes;
        // (a scratch in graded pieces is already placed by its grade: ensure_chain)
        while (d->chain_pieces.empty() && held.size() < candidates && room_for(bytes)) {
            void *fresh = nullptr;
            if (bsgs_big_malloc(&fresh, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            d->chain = (u32x4 *)fresh;
            held.push_back((u32x4 *)fresh);
            if ((rc = timed(&t))) break;
            ms.push_back(t);
        }
        size_t best = 0;
        for (size_t k = 1; k < ms.size(); k++) if (ms[k] < ms[best] * 0.995f) best = k;      // a new placement has to win by 0.5 %
        (void)hipStreamSynchronize(d->stream);
        for (size_t k = 0; k < held.size(); k++) if (k != best && held[k]) (void)hipFree(held[k]);
        d->chain = held[best];
        if (rc) return rc;
        if (ms_out) for (size_t k = 0; k < ms.size(); k++) ms_out[k] = ms[k];
        if (chosen) chosen[0] = (uint32_t)best;
        best_ms = ms[best];
    }
    // ---- bucket lines (only the engine's own copy can move)
    if (chosen) chosen[1] = 0;
    // Not when the scratch lies in graded pieces -- they were graded AGAINST these very lines (alloc_graded_pieces): moving the lines would
    // make every grade stale and could undo a reserved-group placement -- and not for tables above 40 GiB (a copy per candidate, and
    // bsgs_lines_malloc already placed them around the reserved group).
    if (bsgs_lines_layout(d) && d->lines && d->lines_owned && d->chain_pieces.empty() && d->lines_bytes <= (40ull << 30)) {
        std::vector<u32x4 *> held;
        std::vector<float> ms;
        held.push_back(d->lines); ms.push_back(best_ms);
        const uint64_t bytes = d->lines_bytes;
        float t = 0.f;
        while (held.size() < candidates && room_for(bytes)) {
            void *fresh = nullptr;
            if (bsgs_big_malloc(&fresh, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            held.push_back((u32x4 *)fresh);
            if (hipMemcpy(fresh, held[0], bytes, hipMemcpyDeviceToDevice) != hipSuccess) { rc = fail(BSGS_ERR_HIP, "copying the bucket lines"); break; }
            d->lines = (u32x4 *)fresh;
            if ((rc = timed(&t))) break;
            ms.push_back(t);
        }
        size_t best = 0;
        for (size_t k = 1; k < ms.size(); k++) if (ms[k] < ms[best] * 0.995f) best = k;
        (void)hipStreamSynchronize(d->stream);
        for (size_t k = 0; k < held.size(); k++) if (k != best) (void)bsgs_big_free(held[k]);
        d->lines = held[best];
        if (rc) return rc;
        if (ms_out) for (size_t k = 0; k < ms.size(); k++) ms_out[candidates + k] = ms[k];
        if (chosen) chosen[1] = (uint32_t)best;
        best_ms = ms[best];
    }
    // ---- let the driver finish wiping what was freed
    // The wipe of the freed buffers comes in bursts of ~0.5 s, up to 1.5 s apart (profiles/r02g_settling_after_tuning.log): the call is
    // over after twelve launches in a row at the chosen time (2 s), 12 s at most.
    float t = 0.f;
    for (int k = 0, calm = 0; k < 72 && calm < 12; k++) {
        if ((rc = launch(&t))) return rc;
        calm = t <= best_ms * 1.015f ? calm + 1 : 0;
    }
    if (final_ms) *final_ms = t;
    return BSGS_OK;
}

// ---- probe digest (parity instrumentation): per engine thread, XOR and wrapping sum of every 64-bit key it probed -------
extern "C" int bsgs_run_digest(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles, uint64_t *digest_out, bsgs_hit_ex *hits,
                               uint32_t max_hits, uint32_t *nhits)
{
    if (!d || !centres || !digest_out) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    if ((d->pi & 1u) || !bsgs_lines_layout(d)) return fail(BSGS_ERR_STATE, "the digest is an instrument of the default (pair-batched, bucket-line) kernel");
    HIPCHK(hipSetDevice(d->id));
    const uint64_t bytes = (uint64_t)ntiles * d->Ti * 16;
    HIPCHK(hipMalloc(&d->digest, bytes));
    hipError_t e = hipMemsetAsync(d->digest, 0, bytes, d->stream);
    const unsigned saved_flags = d->debug_flags;
    const int saved_variant = d->variant;
    d->debug_flags = 8u; d->variant = 13; d->phase_probe = true;       // the default kernel (quad chain; pair chain for odd batch lengths), instrumented instantiation
    int rc = e == hipSuccess ? bsgs_run(d, centres, ntiles, hits, max_hits, nhits, nullptr) : fail(BSGS_ERR_HIP, "memset: %s", hipGetErrorString(e));
    d->debug_flags = saved_flags; d->variant = saved_variant; d->phase_probe = false;
    if (rc == BSGS_OK || rc == BSGS_ERR_OVERFLOW) {
        e = hipMemcpy(digest_out, d->digest, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(BSGS_ERR_HIP, "digest read-back: %s", hipGetErrorString(e));
    }
    (void)hipFree(d->digest);
    d->digest = nullptr;
    return rc;
}

// diagnostics: one walk launch of `ntiles` tiles with every block recording its XCD; out[2x] = time (100 MHz ticks, relative to the
// earliest XCD's last block) at which XCD x finished its last block, out[2x+1] = blocks XCD x ran.  The block -> XCD assignment is
// static (blockIdx % 8): an XCD that runs slower than the others (per-XCD clocks under the power cap) sets the launch time.
static __global__ void wallclock_kernel(unsigned long long *out) { out[0] = wall_clock64(); }
extern "C" int bsgs_debug_xcd_profile(bsgs_dev *d, uint64_t first_tile, uint32_t ntiles, uint64_t out[16], float *launch_ms)
{
    if (!d || !out) return fail(BSGS_ERR_ARG, "null");
    if (!d->walk_set) return fail(BSGS_ERR_STATE, "bsgs_set_walk first");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued");
    if ((d->pi & 1u) || !bsgs_lines_layout(d)) return fail(BSGS_ERR_STATE, "default kernel only");
    HIPCHK(hipSetDevice(d->id));
    HIPCHK(hipMalloc(&d->digest, 17 * 8));
    hipError_t e = hipMemsetAsync(d->digest, 0, 17 * 8, d->stream);
    hipLaunchKernelGGL(wallclock_kernel, dim3(1), dim3(1), 0, d->stream, (unsigned long long *)d->digest + 16);
    const unsigned saved_flags = d->debug_flags;
    const int saved_variant = d->variant;
    const uint32_t saved_tpl = d->tiles_per_launch;
    d->debug_flags = 16u; d->variant = 13; d->phase_probe = true; d->tiles_per_launch = ntiles;
    // launch_tiles offsets the digest pointer by seq * Ti * 2: one launch, seq = 0
    int rc = e == hipSuccess ? bsgs_run_walk(d, first_tile, ntiles, nullptr, 0, nullptr, launch_ms) : fail(BSGS_ERR_HIP, "memset");
    d->debug_flags = saved_flags; d->variant = saved_variant; d->phase_probe = false; d->tiles_per_launch = saved_tpl;
    uint64_t h[17];
    if (rc == BSGS_OK || rc == BSGS_ERR_OVERFLOW) {
        rc = BSGS_OK;
        if (hipMemcpy(h, d->digest, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(BSGS_ERR_HIP, "read-back");
        else for (int x = 0; x < 8; x++) { out[2 * x] = h[2 * x] ? h[2 * x] - h[16] : 0; out[2 * x + 1] = h[2 * x + 1]; }
    }
    (void)hipFree(d->digest);
    d->digest = nullptr;
    return rc;
}

// ---- phase timing: the same batch run with the kernel stopping after phase 1, after phase 2, and in full ------
extern "C" int bsgs_profile_phases(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles, float ms_out[3])
{
    if (!d || !centres || !ms_out) return fail(BSGS_ERR_ARG, "null");
    if (bsgs_chain_group(d, d->pi) < 2) return fail(BSGS_ERR_STATE, "phase timing is an instrument of the chained kernel (bucket lines, even batch length)");
    const unsigned saved_flags = d->debug_flags;
    const unsigned flags[3] = {1u, 2u, 0u};
    int rc = BSGS_OK;
    d->phase_probe = true;
    for (int k = 0; k < 3 && rc == BSGS_OK; k++) {
        d->debug_flags = flags[k];
        for (int rep = 0; rep < 2 && rc == BSGS_OK; rep++) {   // first repetition warms up
            uint32_t nh = 0;
            rc = bsgs_run(d, centres, ntiles, nullptr, 0, &nh, &ms_out[k]);
            if (rc == BSGS_ERR_OVERFLOW) rc = BSGS_OK;
        }
    }
    d->debug_flags = saved_flags;
    d->phase_probe = false;
    return rc;
}

// ---- selftests ----------------------------------------------------------------------------------------------
extern "C" int bsgs_selftest_fe(bsgs_dev *d, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, uint32_t n)
{
    if (!d || !a || !b || !out) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    fe *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc(&da, (size_t)n * 32)); HIPCHK(hipMalloc(&db, (size_t)n * 32)); HIPCHK(hipMalloc(&dout, (size_t)n * 32));
    HIPCHK(hipMemcpy(da, a, (size_t)n * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, (size_t)n * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fe_selftest_kernel, dim3((n + 63) / 64), dim3(64), 0, d->stream, op, da, db, dout, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 32, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "selftest_fe: %s", hipGetErrorString(e));
    return BSGS_OK;
}

// the low-64-bit squaring path against the full-width one on n*iters pseudo-random cases: counts[0] mismatches (must be 0),
// counts[1] cases that took the exact path, counts[2] cases
extern "C" int bsgs_selftest_lo64(bsgs_dev *d, const uint8_t *a, const uint8_t *b, uint32_t n, uint32_t iters, uint64_t counts[3])
{
    if (!d || !a || !b || !counts || !n) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    fe *da = nullptr, *db = nullptr; unsigned long long *dc = nullptr;
    HIPCHK(hipMalloc(&da, (size_t)n * 32)); HIPCHK(hipMalloc(&db, (size_t)n * 32)); HIPCHK(hipMalloc(&dc, 24));
    HIPCHK(hipMemcpy(da, a, (size_t)n * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, b, (size_t)n * 32, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dc, 0, 24));
    hipLaunchKernelGGL(lo64_selftest_kernel, dim3((n + 255) / 256), dim3(256), 0, d->stream, da, db, dc, n, iters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "selftest_lo64: %s", hipGetErrorString(e));
    counts[0] = h[0]; counts[1] = h[1]; counts[2] = h[2];
    return BSGS_OK;
}

extern "C" int bsgs_selftest_xs(bsgs_dev *d, const uint8_t px_le[32], const uint8_t py_le[32], uint64_t first, uint32_t count, uint8_t *out)
{
    if (!d || !px_le || !py_le || !out) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants");
    if (first + count > d->maxnonce) return fail(BSGS_ERR_ARG, "range beyond maxnonce");
    HIPCHK(hipSetDevice(d->id));
    fe *dout = nullptr;
    HIPCHK(hipMalloc(&dout, (size_t)count * 96));
    fe Px, Py;
    bsgs_le_to_fe(Px, px_le); bsgs_le_to_fe(Py, py_le);
    hipLaunchKernelGGL(xs_selftest_kernel, dim3((count + 63) / 64), dim3(64), 0, d->stream, d->g2, d->Ti, d->pi, Px, Py, first, count, dout);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)count * 96, hipMemcpyDeviceToHost);
    (void)hipFree(dout);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "selftest_xs: %s", hipGetErrorString(e));
    return BSGS_OK;
}

// ---- roofline denominators ---------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mb_splitmix(u64 &s)
{
    s += 0x9E3779B97F4A7C15ULL;
    u64 z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <int LP>
__global__ void __launch_bounds__(256) mb_gups_kernel(const u32x4 *__restrict__ buf, u64 n_gran_mask, int iters, u32 *out, u64 seed)
{
    const u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
    u64 s = seed + (u64)(tid / LP) * 0x632BE59BD9B4E019ULL;
    const u32 sub = tid % LP;
    u32 acc = 0;
    for (int i = 0; i < iters; i++) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = buf[(mb_splitmix(s) & n_gran_mask) * LP + sub];
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}

extern "C" int bsgs_bench_random_read(bsgs_dev *d, uint64_t footprint_bytes, uint32_t granule, double *gbps, double *greads)
{
    if (!d || (granule != 32 && granule != 64 && granule != 128)) return fail(BSGS_ERR_ARG, "granule must be 32, 64 or 128");
    HIPCHK(hipSetDevice(d->id));
    uint64_t n = 1;
    while (n * 2 * granule <= footprint_bytes) n *= 2;         // power-of-two granule count
    void *buf = nullptr; u32 *out = nullptr;
    HIPCHK(hipMalloc(&buf, n * granule));
    HIPCHK(hipMalloc(&out, 64));
    HIPCHK(hipMemsetAsync(buf, 0x5a, n * granule, d->stream));
    const int blocks = 256 * 8, iters = 256;
    const int LP = (int)granule / 16;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        HIPCHK(hipEventRecord(e0, d->stream));
        if (LP == 4) hipLaunchKernelGGL(mb_gups_kernel<4>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)buf, n - 1, iters, out, 17ull + rep);
        else if (LP == 2) hipLaunchKernelGGL(mb_gups_kernel<2>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)buf, n - 1, iters, out, 17ull + rep);
        else         hipLaunchKernelGGL(mb_gups_kernel<8>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)buf, n - 1, iters, out, 17ull + rep);
        HIPCHK(hipEventRecord(e1, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double reads = (double)blocks * 256 * iters * 8 / LP;
    if (greads) *greads = reads / (ms * 1e-3) / 1e9;
    if (gbps) *gbps = reads * granule / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(out);
    return BSGS_OK;
}

// ---- counter calibration streams -----------------------------------------------------------------------------------------------
// rocprofv3's FETCH_SIZE / WRITE_SIZE are request counters with a nominal size; what they report per byte depends on the access pattern
// (MI355X_MICROARCH.md, HBM: wide coalesced 16-byte-per-lane reads are tallied at 1/2).  The tile kernel mixes three patterns -- random
// 4 x 16-byte line reads by LDS-DMA (the probes), coalesced 16-byte-per-lane reads by plain loads and by LDS-DMA (giants, stored products),
// coalesced non-temporal 16-byte stores (stored products) -- so bench.py's counter passes run each pattern ONCE over a known number of bytes
// in the same process and divide: kind 0 = plain coalesced reads, 1 = coalesced reads by global_load_lds_dwordx4, 2 = non-temporal stores;
// bsgs_bench_random_read is the probe pattern.  Every kernel touches each of the `bytes` exactly once.
static __global__ void __launch_bounds__(256) mb_stream_read_kernel(const u32x4 *__restrict__ buf, u64 n16, u32 *out)
{
    u32 acc = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        const u32x4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}
static __global__ void __launch_bounds__(256) mb_stream_read_lds_kernel(const u32x4 *__restrict__ buf, u64 n16, u32 *out)
{
    __shared__ __attribute__((aligned(16))) char slot[4096];                     // 1 KiB per wave: where the DMA lands
    const u32 wave_base = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024u);
    u32 acc = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < ((n16 + 63) & ~63ull); i += (u64)gridDim.x * blockDim.x) {
        if (i < n16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(buf + i),
                                                      (__attribute__((address_space(3))) void *)(slot + wave_base), 16, 0, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc ^= *(const u32 *)(slot + wave_base + (threadIdx.x & 63) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x9abcdef1u) out[0] = acc;
}
static __global__ void __launch_bounds__(256) mb_stream_write_nt_kernel(u32x4 *__restrict__ buf, u64 n16, u32 seed)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n16; i += (u64)gridDim.x * blockDim.x) {
        const u32x4 v = {seed, (u32)i, (u32)(i >> 32), seed ^ (u32)i};
        __builtin_nontemporal_store(v, buf + i);
    }
}
extern "C" int bsgs_bench_stream(bsgs_dev *d, int kind, uint64_t bytes, double *gbps)
{
    if (!d || kind < 0 || kind > 2 || bytes < (1ull << 20)) return fail(BSGS_ERR_ARG, "kind 0..2, at least 1 MiB");
    HIPCHK(hipSetDevice(d->id));
    void *buf = nullptr; u32 *out = nullptr;
    HIPCHK(bsgs_big_malloc(&buf, bytes));
    if (hipMalloc(&out, 64) != hipSuccess) { (void)hipFree(buf); return fail(BSGS_ERR_NOMEM, "out word"); }
    hipError_t e = hipMemsetAsync(buf, 0x5a, bytes, d->stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    const u64 n16 = bytes / 16;
    const int blocks = d->prop.multiProcessorCount * 16;
    if (e == hipSuccess) e = hipEventRecord(e0, d->stream);
    if (kind == 0) hipLaunchKernelGGL(mb_stream_read_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)buf, n16, out);
    else if (kind == 1) hipLaunchKernelGGL(mb_stream_read_lds_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)buf, n16, out);
    else hipLaunchKernelGGL(mb_stream_write_nt_kernel, dim3(blocks), dim3(256), 0, d->stream, (u32x4 *)buf, n16, 7u);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf); (void)hipFree(out);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "bench_stream: %s", hipGetErrorString(e));
    if (gbps) *gbps = ms > 0.f ? bytes / (ms * 1e-3) / 1e9 : 0.0;
    return BSGS_OK;
}

// diagnostics: where the engine's buffers live (device virtual addresses: lines, chain, giants, csr, centres) and how fast the
// installed bucket lines THEMSELVES can be read at random (the same cooperative 4-lane pattern as the probe) -- the physical
// placement of these buffers moves the launch time by up to 10 % (tools/placement_probe.py)
extern "C" int bsgs_debug_buffers(bsgs_dev *d, uint64_t addr[5], double *lines_random_read_gbps)
{
    if (!d || !addr) return fail(BSGS_ERR_ARG, "null");
    addr[0] = (uint64_t)d->lines; addr[1] = (uint64_t)(d->chain_pieces.empty() ? d->chain : d->chain_pieces[0]); addr[2] = (uint64_t)d->g2; addr[3] = (uint64_t)d->csr; addr[4] = (uint64_t)d->cen_dev;
    if (lines_random_read_gbps) {
        *lines_random_read_gbps = 0;
        if (d->lines && d->layout == BSGS_TABLE_LINES64) {
            HIPCHK(hipSetDevice(d->id));
            uint64_t n = 1;
            while (n * 2 <= d->ht_items) n *= 2;
            u32 *out = nullptr;
            HIPCHK(hipMalloc(&out, 64));
            hipEvent_t e0, e1;
            HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
            const int blocks = 256 * 8, iters = 128;
            for (int rep = 0; rep < 2; rep++) {
                HIPCHK(hipEventRecord(e0, d->stream));
                hipLaunchKernelGGL(mb_gups_kernel<4>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->lines, n - 1, iters, out, 91ull + rep);
                HIPCHK(hipEventRecord(e1, d->stream));
                HIPCHK(hipStreamSynchronize(d->stream));
            }
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            *lines_random_read_gbps = (double)blocks * 256 * iters * 8 / 4 * 64 / (ms * 1e-3) / 1e9;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(out);
        }
    }
    return BSGS_OK;
}

__global__ void __launch_bounds__(256) mb_modmul_kernel(fe *out, int iters, u32 seed)
{
    const u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    fe a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a.v[i] = seed * 2654435761u + t * 40503u + i; b.v[i] = a.v[i] ^ 0x9E3779B9u; }
    for (int i = 0; i < iters; i++) { fe_mul(a, a, b); fe_mul(b, b, a); }
    if (a.v[0] == 0x12345678u && b.v[3] == 7u) out[t] = a;
}

extern "C" int bsgs_bench_modmul(bsgs_dev *d, double *gmul)
{
    if (!d || !gmul) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    const int blocks = 256 * 8, iters = 2000;
    fe *out = nullptr;
    HIPCHK(hipMalloc(&out, (size_t)blocks * 256 * 32));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(mb_modmul_kernel, dim3(blocks), dim3(256), 0, d->stream, out, 10, 1u);
    HIPCHK(hipEventRecord(e0, d->stream));
    hipLaunchKernelGGL(mb_modmul_kernel, dim3(blocks), dim3(256), 0, d->stream, out, iters, 2u);
    HIPCHK(hipEventRecord(e1, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *gmul = (double)blocks * 256 * iters * 2 / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return BSGS_OK;
}
