// baby_builder.hip -- GPU baby-step table builder.
//
// Replaces the reference's CPU pipeline GenBabys -> baby() -> HashTableInsert -> sortWholeHashTable ->
// packHTFile / packHTGPUFile (1_9_7File.pb:1237-1328, 1162-1235, 2555-2622, 2771-2895, 3232-3444) -- hours for
// -w 30 on CPU threads serialised by one table mutex -- by:
//   1. baby_keys_kernel: k*G for k = 1..w with the tile kernel's batched-inverse structure
//      (thread tid walks S_tid + j*(T*G), S_tid = (first+tid)*G), emitting key64 = x_le[0:8] in position order;
//   2. rocPRIM device radix sort of (bucket << 32 | hash) with the positions as values (stable, so entries with
//      an identical (bucket, hash) pair stay in ascending position order, the oracle's convention);
//   3. csr_* kernels: bucket starts + the two file images (htGPU: hashes, htCPU: {hash, position}).
// The images are byte-identical to the reference's files for the same (w, htsz) (tests compare with the oracle
// and with the sha256 digests of tests/golden/cfg1_digests.json).
#include "bsgs_internal.h"
#include "support_kernels.hip.h"
#include "host_secp.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

// Point generation: thread tid walks S_tid + j*(T*G), S_tid = (first + tid)*G, j = 0..pi-1, with the tile kernel's batched inverse over its pi
// points, and hands the 64-bit key (low 64 bits of x) of point number first + j*T + tid to a SINK:
//   SINK 0: keys[j*T + tid] = bucket << 32 | hash, pos[j*T + tid] = its position (the reference-format images need the positions: sorted next);
//   SINK 2 / 3: straight into 64 / 128-byte bucket lines (one claim-a-slot atomic per key, fused into the generator: no key array, and the memory-bound
//   scatter overlaps with the arithmetic of the other waves).  The slot an atomic returns is used one point LATER, so the wave never waits for it.
// Only indices < count are produced.  helper[j-1] = j*(T*G) (uniform: scalar loads); chain = [pi][2][T] scratch.
struct KeySink {
    u64 *keys; u32 *pos; u32 pos_base;  // SINK 0: sort key (bucket << 32 | hash) and position of point number idx (197:2561, 2583, 1221)
    u32 *lines; u64 *ovf; u64 ovf_cap; unsigned long long *counters; u32 mask;      // SINK 2 / 3: the line of bucket b counts its arrivals in word 0
    u32 mul;                            // 0: bucket = x & mask ; M: any number of buckets, bucket from 48 bits of the key (giant_kernel.hip.h bucket_mul48)
    u64 region;                         // SINK 2 / 3: the overflow list is filled in one REGION of this many entries per block of the generator, each with a counter of its own
                                        // (counters[16 + 16 * block]: 128 bytes apart).  One counter for everybody held the scatter of the large tables 1.5-2.3 s above its
                                        // random-write bound: 2 * 10^9 appends at 36 * 2^30 points, every one through the same address (profiles/r08s_*)
    u64 tail_base, tail_cap;            // SINK 2 / 3: a block whose region is full appends to the shared TAIL of the list (entries [tail_base, tail_base + tail_cap), one counter: counters[8]) --
                                        // a region is 1/blocks of the list, and blocks that run late (a GPU that does not hold all of them at once, a single-chunk table) see fuller lines
    u32 b_lo, b_hi;                     // SINK 2 / 3: only buckets [b_lo, b_hi) are filed, in lines[(bucket - b_lo) * WORDS] (a SLICE of the table: one engine of N builds 1/N of the
                                        // lines -- every engine generates every point -- and an all-gather completes them; the whole table: 0, number of buckets)
};
template <int SINK>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) baby_keys_kernel(const u32x4 *__restrict__ helper, const u32x4 *__restrict__ bases, const KeySink K,
                                                        u32x4 *__restrict__ chain, u32 T, u32 pi, u64 count)
{
    constexpr u32 WORDS = SINK == 3 ? 32u : 16u, CAP = WORDS - 1;
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= T) return;
    // the claim whose slot number is still in flight: three registers (bucket, hash, slot) -- the line's address is recomputed from the bucket when the claim is
    // settled (carrying the pointer and a 64-bit bucket cost the SINK 2 / 3 instantiations three spilled VGPRs at their 128-register budget)
    constexpr u32 NONE = 0xFFFFFFFFu;  // never a bucket: buckets are < M < 2^32
    u32 pend_bucket = NONE, pend_slot = 0, pend_hash = 0;
    auto settle = [&]() {
        if (SINK == 0 || pend_bucket == NONE) return;
        if (pend_slot < CAP - 1) K.lines[(u64)(pend_bucket - K.b_lo) * WORDS + 1 + pend_slot] = pend_hash;     // CAP - 1 arrivals in the line; word CAP is the bound (ext_refine_kernel)
        else {
            const u64 at = atomicAdd(K.counters + 16 + (u64)blockIdx.x * 16, 1ull);
            if (at < K.region) K.ovf[(u64)blockIdx.x * K.region + at] = ((u64)pend_bucket << 32) | pend_hash;
            else {
                const u64 tl = atomicAdd(K.counters + 8, 1ull);               // (an entry beyond the tail is counted, not written: the host sees the count and refuses the table)
                if (tl < K.tail_cap) K.ovf[K.tail_base + tl] = ((u64)pend_bucket << 32) | pend_hash;
            }
        }
        pend_bucket = NONE;
    };
    auto emit = [&](u64 idx, u64 key) {
        if (idx >= count) return;
        if (SINK == 0) { K.keys[idx] = ((u64)((u32)key & K.mask) << 32) | (key >> 32); K.pos[idx] = K.pos_base + (u32)idx; return; }
        settle();
        const u32 bucket = K.mul ? bucket_mul48((u32)key, (u32)(key >> 32), K.mul) : ((u32)key & K.mask);
        if (bucket < K.b_lo || bucket >= K.b_hi) return;                       // another engine's slice
        pend_bucket = bucket;
        pend_hash = (u32)(key >> 32);
        pend_slot = atomicAdd(K.lines + (u64)(bucket - K.b_lo) * WORDS, 1u);
    };
    fe Sx, Sy;
    fe_load2(Sx, bases + (u64)tid * 4 + 0, bases + (u64)tid * 4 + 1);
    fe_load2(Sy, bases + (u64)tid * 4 + 2, bases + (u64)tid * 4 + 3);
    emit(tid, ((u64)Sx.v[1] << 32) | Sx.v[0]);
    if (pi == 1) { settle(); return; }
    fe nSx;
    fe_neg(nSx, Sx);
    fe acc;
    fe_set_one(acc);
    for (u32 j = 1; j < pi; j++) {
        fe hx, d;
        fe_load2(hx, helper + (u64)(j - 1) * 4 + 0, helper + (u64)(j - 1) * 4 + 1);
        fe_sub(d, hx, Sx);
        if (__builtin_expect(fe_eq(hx, Sx), 0)) fe_add(d, Sy, Sy);          // S + S: tangent
        fe_mul(acc, acc, d);
        fe_store2_nt(chain + ((u64)j * 2 + 0) * T + tid, chain + ((u64)j * 2 + 1) * T + tid, acc);
    }
    fe inv;
    fe_inv(inv, acc);
    fe c;                                                                    // the running product before point j, requested one point ahead
    if (pi > 2) fe_load2_nt(c, chain + ((u64)(pi - 2) * 2 + 0) * T + tid, chain + ((u64)(pi - 2) * 2 + 1) * T + tid);
    for (u32 j = pi - 1; j >= 1; j--) {
        fe hx, hy, d, s, t, lam, nhx;
        fe_load2(hx, helper + (u64)(j - 1) * 4 + 0, helper + (u64)(j - 1) * 4 + 1);
        fe_load2(hy, helper + (u64)(j - 1) * 4 + 2, helper + (u64)(j - 1) * 4 + 3);
        const bool dbl = fe_eq(hx, Sx);
        fe_sub(d, hx, Sx);
        if (__builtin_expect(dbl, 0)) fe_add(d, Sy, Sy);
        if (j > 1) {
            fe_mul(s, inv, c);
            fe_mul(inv, inv, d);
            if (j > 2) fe_load2_nt(c, chain + ((u64)(j - 2) * 2 + 0) * T + tid, chain + ((u64)(j - 2) * 2 + 1) * T + tid);
        } else {
            s = inv;
        }
        fe_sub(t, hy, Sy);
        if (__builtin_expect(dbl, 0)) { fe x2; fe_sqr(x2, Sx); fe_add(t, x2, x2); fe_add(t, t, x2); }
        fe_mul(lam, t, s);
        fe_neg(nhx, hx);
        fe_lo64_addends cad;
        fe_lo64_prepare(cad, nSx, nhx);                                      // the probe reads 64 bits of x: so does the table (fp256.hip.h)
        emit((u64)j * T + tid, x_key_from_lambda(lam, nSx, nhx, cad));
    }
    settle();
}

// bucket starts: off[b] = number of entries in buckets < b ; off[ht_items] = w
__global__ void csr_offsets_kernel(const u64 *__restrict__ sk, u32 *__restrict__ off, u64 n, u64 ht_items)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i <= n; i += (u64)gridDim.x * blockDim.x) {
        const u64 b_prev = i == 0 ? 0 : (sk[i - 1] >> 32) + 1;              // first bucket not yet closed before entry i
        const u64 b_cur = i == n ? ht_items : (sk[i] >> 32);
        for (u64 b = b_prev; b <= b_cur; b++) off[b] = (u32)i;
    }
}

__global__ void csr_items_kernel(const u64 *__restrict__ sk, const u32 *__restrict__ pos, u32 *__restrict__ gpu_items,
                                 u32 *__restrict__ cpu_items, u64 n)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 h = (u32)sk[i];
        if (gpu_items) gpu_items[i] = h;
        if (cpu_items) { cpu_items[2 * i] = h; cpu_items[2 * i + 1] = pos[i]; }
    }
}

// every point of the reference-format build left its position behind: pos[i] == i for all i < n (the array is pre-filled with 0xFFFFFFFF, never a position).  A point the
// generator did not produce -- which the sort would file as a garbage entry among w, invisible to a census and to all but a lucky sample -- is counted here
__global__ void __launch_bounds__(256) positions_written_kernel(const u32 *__restrict__ pos, u64 n, unsigned long long *bad)
{
    unsigned long long miss = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) miss += pos[i] != (u32)i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) miss += __shfl_xor(miss, o);
    if ((threadIdx.x & 63) == 0 && miss) atomicAdd(bad, miss);
}

namespace {
// BSGS_BUILD_VERBOSE=1: stage times of the builders on stderr (wall clock; `sync` drains the stream first so that a stage owns its GPU time)
struct StageClock {
    bool on = getenv("BSGS_BUILD_VERBOSE") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(bsgs_dev *d, const char *what, bool sync = true)
    {
        if (!on) return;
        if (sync) (void)hipStreamSynchronize(d->stream);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[build] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return bsgs_big_malloc(&p, n ? n : 1); }      // parked scratch pieces are handed back on demand
    template <class T> T *as() { return (T *)p; }
};

// The point generator shared by both builders: k*G for k = first .. first + count - 1, `chunk` points per launch, nothing on the host in between
// (round 3 computed every chunk's 65536 base points on the host, one synchronisation per chunk, and ran the kernel with one wave per SIMD).
//   bases (first + tid)*G: walk_centres_kernel -- the device walk of the tile centres -- with D = G and P0 = first*G;
//   helper j*(T*G), j = 1..pi-1: host EC library, once.
struct KeyGen {
    bsgs_dev *d = nullptr;
    uint32_t T = 0, pi = 0;
    uint64_t chunk = 0;
    DevBuf chainb, helperb, basesb, gtab, status;
    static void geometry(uint64_t w, uint32_t &T, uint32_t &pi)
    {
        // points per inversion (270 multiplications per thread): 1024 for the tables that take seconds to build; below 2^31 points the build is a tenth of a second of
        // arithmetic and what counts is the scratch it allocates (32 bytes per point of a chunk: 8 GiB at 1024, 2 GiB at 256 -- the driver clears what it hands out at 45 GB/s)
        pi = w >= (1ull << 31) ? 1024 : 256;
        uint64_t t = 256;
        while (t < (1u << 18) && t * pi < w) t *= 2;                // 2^18 threads = four waves per SIMD on 256 CUs; fewer for small tables
        T = (uint32_t)t;
    }
    static uint64_t scratch_bytes(uint64_t w) { uint32_t T, pi; geometry(w, T, pi); return (uint64_t)T * pi * 32 + (uint64_t)T * 64 + (1u << 20); }
    int init(bsgs_dev *dev, uint64_t w)
    {
        d = dev;
        geometry(w, T, pi);
        chunk = (uint64_t)T * pi;
        HIPCHK(chainb.alloc((uint64_t)T * pi * 32));
        HIPCHK(basesb.alloc((size_t)T * 64));
        HIPCHK(status.alloc(4));
        HIPCHK(hipMemsetAsync(status.p, 0, 4, d->stream));
        const hs::Affine TG = hs::point_mul(hs::G, hs::fe_from_u64(T));
        std::vector<hs::Affine> helper = hs::multiples(TG, pi - 1);
        std::vector<uint8_t> hb((size_t)(pi - 1) * 64), tab(64 * 64);
        for (size_t i = 0; i + 1 < pi; i++) hs::affine_to_le(helper[i], &hb[i * 64], &hb[i * 64 + 32]);
        hs::Affine cur = hs::G;
        for (int j = 0; j < 64; j++) { hs::affine_to_le(cur, &tab[(size_t)j * 64], &tab[(size_t)j * 64 + 32]); cur = hs::point_add(cur, cur); }
        HIPCHK(helperb.alloc(hb.size()));
        HIPCHK(gtab.alloc(tab.size()));
        HIPCHK(hipMemcpy(helperb.p, hb.data(), hb.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(gtab.p, tab.data(), tab.size(), hipMemcpyHostToDevice));
        return BSGS_OK;
    }
    // after the last run() and a stream synchronisation: did the device walk produce every base point?  (an infinity among the bases (first + tid)*G would be
    // filed as a dummy point; unreachable for k < 2^36 < n, but never silently)
    int check()
    {
        uint32_t bad = 0;
        HIPCHK(hipMemcpy(&bad, status.p, 4, hipMemcpyDeviceToHost));
        if (bad) return fail(BSGS_ERR_DEGENERATE, "table build: %u base point(s) of the generator are the point at infinity", bad);
        return BSGS_OK;
    }
    // queue the generation of points first .. first + count - 1 (count <= chunk) into the sink; asynchronous
    template <int SINK>
    int run(uint64_t first, uint64_t count, const KeySink &K)
    {
        const hs::Affine start = hs::point_mul(hs::G, hs::fe_from_u64(first));
        fe p0x, p0y;
        hs::affine_to_le(start, (uint8_t *)p0x.v, (uint8_t *)p0y.v);
        hipLaunchKernelGGL(walk_centres_kernel, dim3((T + 63) / 64), dim3(64), 0, d->stream, p0x, p0y, gtab.as<const fe>(), (u64)0, (u32)T,
                           basesb.as<fe>(), status.as<u32>());
        hipLaunchKernelGGL(baby_keys_kernel<SINK>, dim3((T + 255) / 256), dim3(256), 0, d->stream, helperb.as<const u32x4>(), basesb.as<const u32x4>(), K,
                           chainb.as<u32x4>(), T, pi, count);
        HIPCHK(hipGetLastError());
        return BSGS_OK;
    }
};
}  // namespace

// core: images are written to DEVICE buffers gpu_img / cpu_img (either may be NULL)
static int build_to_device(bsgs_dev *d, uint64_t w, uint32_t htsz, u32 *gpu_img, u32 *cpu_img)
{
    const uint64_t ht_items = 1ull << htsz;
    StageClock clk;
    DevBuf sk, sk2, pos, pos2, tmp, offs;
    {   // keys + positions twice over (the sort's output) + the generator's scratch: say so instead of failing half-way
        size_t fr = 0, tot = 0;
        HIPCHK(bsgs_mem_available(&fr, &tot));
        const uint64_t need = 24 * w + KeyGen::scratch_bytes(w) + (64ull << 20);
        if (need > fr) return fail(BSGS_ERR_NOMEM, "table build needs %.1f GiB of device memory, %.1f GiB free", need / 1073741824.0, fr / 1073741824.0);
    }
    HIPCHK(sk.alloc(w * 8)); HIPCHK(pos.alloc(w * 4));
    HIPCHK(hipMemsetAsync(pos.p, 0xFF, w * 4, d->stream));            // "no point yet": positions_written_kernel below
    {
        KeyGen gen;
        int rc = gen.init(d, w);
        if (rc) return rc;
        KeySink K{};
        K.mask = (u32)(ht_items - 1);
        for (uint64_t first = 1; first <= w; first += gen.chunk) {
            K.keys = sk.as<u64>() + (first - 1); K.pos = pos.as<u32>() + (first - 1); K.pos_base = (u32)(first - 1);
            rc = gen.run<0>(first, std::min<uint64_t>(gen.chunk, w - first + 1), K);
            if (rc) return rc;
        }
        HIPCHK(hipStreamSynchronize(d->stream));                 // the generator's scratch goes out of scope here
        rc = gen.check();
        if (rc) return rc;
        DevBuf badb;
        unsigned long long bad = 0;
        HIPCHK(badb.alloc(8));
        HIPCHK(hipMemsetAsync(badb.p, 0, 8, d->stream));
        hipLaunchKernelGGL(positions_written_kernel, dim3((unsigned)std::min<uint64_t>((w + 255) / 256, (uint64_t)d->prop.multiProcessorCount * 16)), dim3(256), 0, d->stream,
                           pos.as<const u32>(), (u64)w, badb.as<unsigned long long>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&bad, badb.p, 8, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (bad) return fail(BSGS_ERR_STATE, "table build: %llu of %llu points were not produced by the generator (their slots still hold the fill pattern)", bad, (unsigned long long)w);
        clk.lap(d, "generate the points (keys)");
    }
    // sort by (bucket, hash), positions ride along (stable: entries with an identical (bucket, hash) pair stay in ascending position order)
    HIPCHK(sk2.alloc(w * 8)); HIPCHK(pos2.alloc(w * 4));
    const int gblocks = (int)std::min<uint64_t>((w + 255) / 256, 1u << 16);
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, sk.as<u64>(), sk2.as<u64>(), pos.as<u32>(), pos2.as<u32>(), (size_t)w, 0u, 32u + htsz, d->stream));
    HIPCHK(tmp.alloc(tmp_bytes));
    HIPCHK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, sk.as<u64>(), sk2.as<u64>(), pos.as<u32>(), pos2.as<u32>(), (size_t)w, 0u, 32u + htsz, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    clk.lap(d, "radix sort by (bucket, hash)");
    (void)hipFree(sk.p); sk.p = nullptr;
    (void)hipFree(pos.p); pos.p = nullptr;
    // images
    const uint64_t hdr = 4 * (ht_items + 1);
    u32 *off_dst = gpu_img ? gpu_img : cpu_img;
    if (!off_dst) { HIPCHK(offs.alloc(hdr)); off_dst = offs.as<u32>(); }
    hipLaunchKernelGGL(csr_offsets_kernel, dim3(gblocks), dim3(256), 0, d->stream, sk2.as<const u64>(), off_dst, w, ht_items);
    hipLaunchKernelGGL(csr_items_kernel, dim3(gblocks), dim3(256), 0, d->stream, sk2.as<const u64>(), pos2.as<const u32>(),
                       gpu_img ? gpu_img + ht_items + 1 : (u32 *)nullptr, cpu_img ? cpu_img + ht_items + 1 : (u32 *)nullptr, w);
    HIPCHK(hipGetLastError());
    if (gpu_img && cpu_img) HIPCHK(hipMemcpyAsync(cpu_img, gpu_img, hdr, hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    clk.lap(d, "bucket starts + file images");
    return BSGS_OK;
}

// expected number of entries beyond `cap` per bucket for Poisson(lambda) loads, times the number of buckets
static double expected_overflow_entries(double lambda, unsigned cap, double buckets)
{
    // E[(X - cap)+] = sum_{c > cap} (c - cap) P(c); P by recurrence in log space
    double e = 0.0;
    const int top = (int)(lambda + 40.0 * std::sqrt(lambda + 1.0) + cap + 64);
    double logp = -lambda;                                      // log P(0)
    for (int c = 1; c <= top; c++) {
        logp += std::log(lambda) - std::log((double)c);
        if ((unsigned)c > cap) e += (double)(c - (int)cap) * std::exp(logp);
    }
    return e * buckets;
}

// `htsz` of the extended-table entry points: 1..31 = 2^htsz buckets (bucket = x & mask, like the reference's tables); a value above 31 IS the number of
// buckets (any number below 2^32: the bucket then comes from 48 bits of the key, giant_kernel.hip.h bucket_mul48) -- what lets a table fill the HBM there is (include/bsgs_hip.h)
static uint64_t ext_buckets(uint32_t htsz) { return htsz <= 31 ? 1ull << htsz : (uint64_t)htsz; }
static uint32_t ext_bucket_mul(uint32_t htsz) { return htsz <= 31 || !(htsz & (htsz - 1)) ? 0u : htsz; }      // the bucket function follows from the bucket COUNT alone: a power of two -> the mask
static int ext_check_args(uint64_t w, uint32_t htsz, uint32_t layout)
{
    if (!w || w > (1ull << 36) || htsz < 1) return fail(BSGS_ERR_ARG, "need 0 < w <= 2^36 and htsz >= 1");
    if (layout != BSGS_TABLE_LINES64_LIST && layout != BSGS_TABLE_LINES128_LIST) return fail(BSGS_ERR_ARG, "layout must be BSGS_TABLE_LINES64_LIST or BSGS_TABLE_LINES128_LIST");
    return BSGS_OK;
}
static int ext_check(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    return ext_check_args(w, htsz, layout);
}

// upper estimate of the entries that will not fit their line (Poisson loads), with slack
static uint64_t ext_list_capacity(uint64_t w, uint32_t htsz, uint32_t layout)
{
    const unsigned cap_line = layout == BSGS_TABLE_LINES128_LIST ? 30 : 14;      // the last word of a full line is the bound of its overflow entries (OVERFLOW BOUND, support_kernels.hip.h)
    const double buckets = (double)ext_buckets(htsz);
    // slack: 25 % for small tables; 5 % once the expectation is large (its relative spread is 1 / sqrt(entries), and the set built from it takes the next power of two above
    // TWICE the capacity: at -w 35 on 3 * 2^30 lines of 64 bytes, 0.95 G expected entries, that is the difference between 16 and 32 GiB)
    const double e = expected_overflow_entries((double)w / buckets, cap_line, buckets);
    return std::min<uint64_t>(w, (uint64_t)((e > 1e8 ? 1.05 : 1.25) * e) + (1u << 20));
}

extern "C" int bsgs_ext_overflow_capacity(uint64_t w, uint32_t htsz, uint32_t layout, uint64_t *cap)
{
    if (!cap) return fail(BSGS_ERR_ARG, "null");
    const int rc = ext_check_args(w, htsz, layout);
    if (rc) return rc;
    *cap = bsgs_ovf_slots(ext_list_capacity(w, htsz, layout));
    return BSGS_OK;
}

// the regions of the overflow list (one per block of the generator, KeySink::region) -> one dense array: region b's `count` entries go to out[offset ...)
static __global__ void __launch_bounds__(256) ovf_compact_kernel(const u64 *__restrict__ list, u64 region, const u64 *__restrict__ count_offset, u64 *__restrict__ out)
{
    const u64 b = blockIdx.y, n = count_offset[2 * b], off = count_offset[2 * b + 1];
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[off + i] = list[b * region + i];
}

// the builder proper, for the buckets [b_lo, b_hi) of the table (the whole table: 0, number of buckets): `lines` = the lines of THOSE buckets (b_hi - b_lo lines, device
// memory), list = room for list_cap overflow entries.  On return the lines are closed and refined and list[0 .. *n_list) holds the overflow entries (bucket << 32 | hash,
// global bucket numbers), sorted; *overflow_buckets = over-full lines of the slice.  The caller makes the hash set (bsgs_ovf_fill) -- of one list, or of the lists of all slices.
// scratch / scratch_bytes: device memory the caller lends for the sort of the overflow list (the whole-table build lends the buffer of the overflow SET, which is filled only
// afterwards: 30 GiB less to ask the driver for at 36 * 2^30 points, and it clears what it hands out); nullptr = allocate.
static int ext_build_lines(bsgs_dev *d, uint64_t w, uint32_t htsz, int lplog, u32x4 *lines, uint64_t b_lo, uint64_t b_hi, u64 *list, uint64_t list_cap, uint64_t *n_list,
                           uint64_t *overflow_buckets, void *scratch = nullptr, size_t scratch_bytes = 0)
{
    u64 *ovf = list;
    const uint64_t ovf_cap = list_cap;
    const uint64_t ht_items = ext_buckets(htsz), line_bytes = 64ull << (lplog - 2), nlines = b_hi - b_lo;
    if (b_lo >= b_hi || b_hi > ht_items) return fail(BSGS_ERR_ARG, "bucket range [%llu, %llu) of %llu", (unsigned long long)b_lo, (unsigned long long)b_hi, (unsigned long long)ht_items);
    StageClock clk;
    size_t fr = 0, tot = 0;
    HIPCHK(bsgs_mem_available(&fr, &tot));
    const uint64_t need = KeyGen::scratch_bytes(w) + (64ull << 20);
    if (need > fr && d->group0_reserve.size() > 8) {                 // the reserve for the chain scratch was sized for a table with a small overflow set: give most of it back
        trim_reserve(d, 8);
        for (int k = 0; k < 100; k++) {                               // the driver wipes what it got back before it hands it out again (about 45 GB/s)
            HIPCHK(bsgs_mem_available(&fr, &tot));
            if (need <= fr) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    }
    if (need > fr) return fail(BSGS_ERR_NOMEM, "extended table build needs %.1f GiB of scratch, %.1f GiB free", need / 1073741824.0, fr / 1073741824.0);
    DevBuf cnt;
    HIPCHK(hipMemsetAsync(lines, 0, nlines * line_bytes, d->stream));
    uint32_t gen_T = 0, gen_pi = 0;
    KeyGen::geometry(w, gen_T, gen_pi);
    // the generator's blocks (at most 1024) each fill a region of the list; 1/16 of the list is a shared tail for the blocks whose region runs full (with equal
    // regions alone a block that sees fuller lines than the average aborted the build although the list as a whole had room; on a small table the blocks' shares differ by several per cent -- blocks that run late meet fuller lines)
    const uint64_t gen_blocks = (gen_T + 255) / 256, region = (ovf_cap - ovf_cap / 16) / gen_blocks, tail_base = region * gen_blocks, tail_cap = ovf_cap - tail_base;
    const size_t cnt_words = 16 + gen_blocks * 16;                                          // [0] over-full lines (ext_finalize_kernel); [8] entries of the tail; [16 + 16 b] entries of block b's region
    HIPCHK(cnt.alloc(cnt_words * 8));
    HIPCHK(hipMemsetAsync(cnt.p, 0, cnt_words * 8, d->stream));
    clk.lap(d, "clear the bucket lines");
    {
        KeyGen gen;
        int rc = gen.init(d, w);
        if (rc) return rc;
        KeySink K{};
        K.lines = (u32 *)lines; K.ovf = ovf; K.ovf_cap = ovf_cap; K.counters = cnt.as<unsigned long long>(); K.mask = (u32)(ht_items - 1); K.mul = ext_bucket_mul(htsz);
        K.region = region; K.tail_base = tail_base; K.tail_cap = tail_cap;
        if (gen.T != gen_T) return fail(BSGS_ERR_STATE, "generator geometry changed under the builder");
        K.b_lo = (u32)b_lo; K.b_hi = (u32)std::min<uint64_t>(b_hi, 0xFFFFFFFFull);
        for (uint64_t first = 1; first <= w && rc == BSGS_OK; first += gen.chunk) {
            const uint64_t count = std::min<uint64_t>(gen.chunk, w - first + 1);
            rc = lplog == 2 ? gen.run<2>(first, count, K) : gen.run<3>(first, count, K);
        }
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(d->stream));                 // the generator's scratch goes out of scope here
        rc = gen.check();
        if (rc) return rc;
        clk.lap(d, "generate + scatter the points");
    }
    const int fblocks = (int)std::min<uint64_t>((nlines + 255) / 256, (uint64_t)d->prop.multiProcessorCount * 32);
    if (lplog == 2) hipLaunchKernelGGL(ext_finalize_kernel<2>, dim3(fblocks), dim3(256), 0, d->stream, lines, nlines, cnt.as<unsigned long long>());
    else            hipLaunchKernelGGL(ext_finalize_kernel<3>, dim3(fblocks), dim3(256), 0, d->stream, lines, nlines, cnt.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    std::vector<unsigned long long> hc(cnt_words);
    HIPCHK(hipMemcpyAsync(hc.data(), cnt.p, cnt_words * 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    h[0] = hc[0];
    std::vector<u64> count_offset(2 * (gen_blocks + 1));             // the regions, then the tail as one more "region" (it starts at gen_blocks * region)
    unsigned long long spilled = 0;
    for (uint64_t b = 0; b < gen_blocks; b++) {
        const unsigned long long cb = hc[16 + 16 * b], nb = std::min<unsigned long long>(cb, region);
        spilled += cb - nb;                                           // what block b appended beyond its region went to the tail
        count_offset[2 * b] = nb; count_offset[2 * b + 1] = h[1];
        h[1] += nb;
    }
    // (an entry beyond the tail was counted, not written: the table would lack it -- never silently)
    if (hc[8] != spilled) return fail(BSGS_ERR_STATE, "overflow list: %llu entries left their regions, the tail counted %llu", spilled, (unsigned long long)hc[8]);
    if (hc[8] > tail_cap) return fail(BSGS_ERR_NOMEM, "overflow list: %llu entries beyond the regions of the generator's blocks, room for %llu in the shared tail (list capacity %llu)",
                                      (unsigned long long)hc[8], (unsigned long long)tail_cap, (unsigned long long)ovf_cap);
    count_offset[2 * gen_blocks] = hc[8]; count_offset[2 * gen_blocks + 1] = h[1];
    h[1] += hc[8];
    if (getenv("BSGS_BUILD_VERBOSE") && hc[8]) fprintf(stderr, "[build] %llu of %llu overflow entries went through the shared tail\n", (unsigned long long)hc[8], h[1]);
    clk.lap(d, "close the lines (pad, count)");
    if (h[1]) {
        // OVERFLOW BOUND (giant_kernel.hip.h): sort the overflow list by (bucket, hash), then per bucket keep the smallest hashes in the line
        // and put the smallest of the others into the line's last word
        // the regions are gathered into one dense array first (`dense`), and the sort writes its output straight back over the list
        DevBuf dense_own, tmp_own, co;
        const size_t dense_bytes = ((size_t)h[1] * 8 + 255) & ~(size_t)255;
        size_t tmp_bytes = 0;
        const unsigned key_bits = 32u + (htsz <= 31 ? htsz : 32u);          // (bucket << 32 | hash): the buckets of a non-power-of-two table need all 32 bits
        HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, (u64 *)nullptr, ovf, (size_t)h[1], 0u, key_bits, d->stream));
        u64 *dense = nullptr;
        void *tmp = nullptr;
        if (scratch && dense_bytes + tmp_bytes <= scratch_bytes) { dense = (u64 *)scratch; tmp = (char *)scratch + dense_bytes; }
        else {
            HIPCHK(dense_own.alloc(dense_bytes));
            HIPCHK(tmp_own.alloc(tmp_bytes));
            dense = dense_own.as<u64>(); tmp = tmp_own.p;
        }
        HIPCHK(co.alloc(count_offset.size() * 8));
        HIPCHK(hipMemcpyAsync(co.p, count_offset.data(), count_offset.size() * 8, hipMemcpyHostToDevice, d->stream));
        hipLaunchKernelGGL(ovf_compact_kernel, dim3(64, (unsigned)gen_blocks + 1), dim3(256), 0, d->stream, (const u64 *)ovf, (u64)region, co.as<const u64>(), dense);
        HIPCHK(hipGetLastError());
        HIPCHK(rocprim::radix_sort_keys(tmp, tmp_bytes, dense, ovf, (size_t)h[1], 0u, key_bits, d->stream));
        const int rblocks = (int)std::min<uint64_t>((h[1] + 255) / 256, 1u << 16);
        if (lplog == 2) hipLaunchKernelGGL(ext_refine_kernel<2>, dim3(rblocks), dim3(256), 0, d->stream, (u32 *)lines, ovf, (u64)h[1], (u64)b_lo);
        else            hipLaunchKernelGGL(ext_refine_kernel<3>, dim3(rblocks), dim3(256), 0, d->stream, (u32 *)lines, ovf, (u64)h[1], (u64)b_lo);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    clk.lap(d, "overflow list: sort, refine");
    *n_list = h[1]; *overflow_buckets = h[0];
    return BSGS_OK;
}

// the whole table: lines = all lines, ovf_table = ovf_slots u64 for the hash set (both device memory)
static int ext_build_into(bsgs_dev *d, uint64_t w, uint32_t htsz, int lplog, u32x4 *lines, u64 *ovf_table, uint64_t ovf_slots, uint64_t *ovf_n, uint64_t *overflow_buckets)
{
    const uint64_t ovf_cap = ovf_slots / 2;                           // the hash set takes at most that many keys
    DevBuf listb;
    HIPCHK(listb.alloc(ovf_cap * 8));
    uint64_t n_list = 0;
    int rc = ext_build_lines(d, w, htsz, lplog, lines, 0, ext_buckets(htsz), listb.as<u64>(), ovf_cap, &n_list, overflow_buckets, ovf_table, (size_t)ovf_slots * 8);
    if (rc) return rc;
    rc = bsgs_ovf_fill(d, listb.as<u64>(), n_list, ovf_table, ovf_slots);
    if (rc) return rc;
    *ovf_n = ovf_slots;
    return BSGS_OK;
}

// One SLICE of an extended table, for the "1/N each + all-gather" start-up of N engines (include/bsgs_hip.h): the lines of the buckets
// [part * M / nparts, (part + 1) * M / nparts) are built IN PLACE inside the full line buffer `lines_dev`, the slice's overflow entries go to list_dev.
extern "C" int bsgs_build_baby_table_ext_slice(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout, void *lines_dev, uint32_t part, uint32_t nparts, void *list_dev,
                                               uint64_t list_cap, uint64_t *n_list, uint64_t *overflow_buckets)
{
    int rc = ext_check(d, w, htsz, layout);
    if (rc) return rc;
    if (!lines_dev || !list_dev || !n_list || !overflow_buckets) return fail(BSGS_ERR_ARG, "null");
    const uint64_t M = ext_buckets(htsz);
    if (!nparts || part >= nparts || M % nparts) return fail(BSGS_ERR_ARG, "part %u of %u: the %llu buckets must divide evenly", part, nparts, (unsigned long long)M);
    HIPCHK(hipSetDevice(d->id));
    const int lplog = layout == BSGS_TABLE_LINES128_LIST ? 3 : 2;
    const uint64_t per = M / nparts, b_lo = per * part;
    return ext_build_lines(d, w, htsz, lplog, (u32x4 *)lines_dev + (b_lo << lplog), b_lo, b_lo + per, (u64 *)list_dev, list_cap, n_list, overflow_buckets);
}
// the overflow hash set of an extended table from its overflow entries (one list, or the concatenated lists of all slices): set_dev = `slots` u64 (bsgs_ext_overflow_capacity)
extern "C" int bsgs_build_overflow_set(bsgs_dev *d, const void *list_dev, uint64_t n, void *set_dev, uint64_t slots)
{
    if (!d || !set_dev || (n && !list_dev)) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    return bsgs_ovf_fill(d, (const u64 *)list_dev, n, (u64 *)set_dev, slots);
}

extern "C" int bsgs_build_baby_table_ext_device(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout, void *lines_dev, void *ovf_dev,
                                                uint64_t ovf_cap, uint64_t *ovf_n, uint64_t *overflow_buckets)
{
    int rc = ext_check(d, w, htsz, layout);
    if (rc) return rc;
    if (!lines_dev || !ovf_dev || !ovf_n || !overflow_buckets) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    return ext_build_into(d, w, htsz, layout == BSGS_TABLE_LINES128_LIST ? 3 : 2, (u32x4 *)lines_dev, (u64 *)ovf_dev, ovf_cap, ovf_n, overflow_buckets);
}

extern "C" int bsgs_install_table_ext_device(bsgs_dev *d, const void *lines_dev, const void *ovf_dev, uint64_t ovf_n, uint64_t overflow_buckets,
                                             uint64_t w, uint32_t htsz, uint32_t layout)
{
    int rc = ext_check(d, w, htsz, layout);
    if (rc) return rc;
    if (!lines_dev || !ovf_dev) return fail(BSGS_ERR_ARG, "null");
    if (ovf_n < 2 || (ovf_n & (ovf_n - 1))) return fail(BSGS_ERR_ARG, "ovf_n must be the slot count returned by the builder (a power of two)");
    HIPCHK(hipSetDevice(d->id));
    const bool mine = d->recv_lines && lines_dev == d->recv_lines && ovf_dev == d->recv_ovf;    // bsgs_alloc_table_ext_recv's buffers
    // (install_lines frees the previous TABLE, never the receive buffers; they change hands only once the install succeeded: a refused table
    // leaves them with bsgs_free_recv / bsgs_dev_close)
    rc = bsgs_install_lines(d, (u32x4 *)lines_dev, layout == BSGS_TABLE_LINES128_LIST ? 3 : 2, (u64 *)ovf_dev, ovf_n, ext_buckets(htsz), w, overflow_buckets);
    if (rc) return rc;
    if (mine) { d->recv_lines = nullptr; d->recv_ovf = nullptr; }
    d->lines_owned = mine;                                            // otherwise borrowed: the caller keeps both buffers alive
    return BSGS_OK;
}

// Receive buffers for an extended table that arrives by broadcast (config 5: rank 0 builds, RCCL broadcasts, every rank installs).  The
// reference uploads its htGPU buffer into memory the per-GPU thread allocated itself (1_9_7File.pb:2251, 2350, 4769-4843); here the buffers
// come from the engine's own allocator, so a table above 40 GiB gets a memory group reserved for the chain scratch exactly as
// bsgs_build_baby_table_ext's does (bsgs_lines_malloc) -- a caller-allocated buffer cannot.  The engine owns both buffers:
// bsgs_install_table_ext_device on these very pointers makes them its table; bsgs_dev_close frees them if they were never installed.
extern "C" int bsgs_alloc_table_ext_recv(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout, void **lines_dev, void **ovf_dev, uint64_t *ovf_cap)
{
    int rc = ext_check(d, w, htsz, layout);
    if (rc) return rc;
    if (!lines_dev || !ovf_dev || !ovf_cap) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    bsgs_free_table(d);
    bsgs_free_recv(d);
    const uint64_t ht_items = ext_buckets(htsz), line_bytes = layout == BSGS_TABLE_LINES128_LIST ? 128 : 64;
    uint64_t cap = 0;
    rc = bsgs_ext_overflow_capacity(w, htsz, layout, &cap);
    if (rc) return rc;
    size_t fr = 0, tot = 0;
    HIPCHK(bsgs_mem_available(&fr, &tot));
    if (ht_items * line_bytes + cap * 8 > fr) return fail(BSGS_ERR_NOMEM, "extended table needs %.1f GiB, %.1f GiB free", (ht_items * line_bytes + cap * 8) / 1073741824.0, fr / 1073741824.0);
    HIPCHK(bsgs_lines_malloc(d, &d->recv_lines, ht_items * line_bytes));
    hipError_t e = bsgs_big_malloc(&d->recv_ovf, cap * 8);
    if (e != hipSuccess) { bsgs_free_recv(d); return fail(BSGS_ERR_HIP, "hipMalloc overflow set: %s", hipGetErrorString(e)); }
    *lines_dev = d->recv_lines; *ovf_dev = d->recv_ovf; *ovf_cap = cap;
    return BSGS_OK;
}

extern "C" int bsgs_build_baby_table_ext(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout)
{
    int rc = ext_check(d, w, htsz, layout);
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->id));
    bsgs_free_table(d);
    const int lplog = layout == BSGS_TABLE_LINES128_LIST ? 3 : 2;
    const uint64_t ht_items = ext_buckets(htsz), line_bytes = 64ull << (lplog - 2);
    uint64_t ovf_cap = 0;
    rc = bsgs_ext_overflow_capacity(w, htsz, layout, &ovf_cap);
    if (rc) return rc;
    size_t fr = 0, tot = 0;
    HIPCHK(bsgs_mem_available(&fr, &tot));
    if (ht_items * line_bytes + ovf_cap * 8 > fr) return fail(BSGS_ERR_NOMEM, "extended table needs %.1f GiB, %.1f GiB free", (ht_items * line_bytes + ovf_cap * 8) / 1073741824.0, fr / 1073741824.0);
    u32x4 *lines = nullptr;
    u64 *ovf = nullptr;
    StageClock clk;
    HIPCHK(bsgs_lines_malloc(d, (void **)&lines, ht_items * line_bytes));
    clk.lap(d, "allocate the bucket lines (placed)");
    hipError_t e = bsgs_big_malloc((void **)&ovf, ovf_cap * 8);
    if (e != hipSuccess) { (void)bsgs_big_free(lines); return fail(BSGS_ERR_HIP, "hipMalloc overflow list: %s", hipGetErrorString(e)); }
    uint64_t n = 0, ob = 0;
    rc = ext_build_into(d, w, htsz, lplog, lines, ovf, ovf_cap, &n, &ob);
    if (rc) { (void)bsgs_big_free(lines); (void)hipFree(ovf); return rc; }
    return bsgs_install_lines(d, lines, lplog, ovf, n, ht_items, w, ob);    // the engine owns both buffers now
}

static int check_args(bsgs_dev *d, uint64_t w, uint32_t htsz)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!w || w >= (1ull << 32) || htsz < 1 || htsz > 31)
        return fail(BSGS_ERR_ARG, "need 0 < w < 2^32 and 1 <= htsz <= 31 (reference limits, 1_9_7File.pb:4412-4418)");
    return BSGS_OK;
}

extern "C" int bsgs_build_baby_tables_device(bsgs_dev *d, uint64_t w, uint32_t htsz, void *htgpu_dev, void *htcpu_dev)
{
    int rc = check_args(d, w, htsz);
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->id));
    return build_to_device(d, w, htsz, (u32 *)htgpu_dev, (u32 *)htcpu_dev);
}

extern "C" int bsgs_build_baby_tables(bsgs_dev *d, uint64_t w, uint32_t htsz, void *htgpu_out, void *htcpu_out, uint32_t install_layout)
{
    int rc = check_args(d, w, htsz);
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->id));
    const uint64_t ht_items = 1ull << htsz, hdr = 4 * (ht_items + 1), gpu_bytes = hdr + 4 * w, cpu_bytes = hdr + 8 * w;
    DevBuf img_gpu, img_cpu;
    const bool want_gpu = htgpu_out != nullptr || install_layout != BSGS_NO_INSTALL;
    if (want_gpu) HIPCHK(img_gpu.alloc(gpu_bytes));
    if (htcpu_out) HIPCHK(img_cpu.alloc(cpu_bytes));
    rc = build_to_device(d, w, htsz, img_gpu.as<u32>(), img_cpu.as<u32>());
    if (rc) return rc;
    if (htgpu_out) HIPCHK(hipMemcpy(htgpu_out, img_gpu.p, gpu_bytes, hipMemcpyDeviceToHost));
    if (htcpu_out) HIPCHK(hipMemcpy(htcpu_out, img_cpu.p, cpu_bytes, hipMemcpyDeviceToHost));
    if (install_layout != BSGS_NO_INSTALL) {
        // hand the device image to the engine (it becomes the owner)
        rc = bsgs_upload_htgpu_device(d, img_gpu.p, ht_items, w, install_layout);
        if (rc) { bsgs_free_table(d); return rc; }        // the borrowed image dies with img_gpu: leave no pointer to it behind
        // layouts that keep the CSR image as their fallback now own it; the *_LIST layouts dropped it (img_gpu frees it)
        if (d->csr == img_gpu.as<u32>()) { d->csr_owned = true; img_gpu.p = nullptr; }
    }
    return BSGS_OK;
}
