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
#include "host_secp.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <vector>

// keys[j*T + tid] = low 64 bits of x((first + j*T + tid) * G); only indices < count are written
__global__ void __launch_bounds__(256) baby_keys_kernel(const u32x4 *__restrict__ helper, const u32x4 *__restrict__ bases,
                                                        u64 *__restrict__ keys, u32x4 *__restrict__ chain, u32 T, u32 pi, u64 count)
{
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= T) return;
    fe Sx, Sy;
    fe_load2(Sx, bases + (u64)tid * 4 + 0, bases + (u64)tid * 4 + 1);
    fe_load2(Sy, bases + (u64)tid * 4 + 2, bases + (u64)tid * 4 + 3);
    if (tid < count) keys[tid] = ((u64)Sx.v[1] << 32) | Sx.v[0];
    if (pi == 1) return;
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
        fe_store2(chain + ((u64)j * 2 + 0) * T + tid, chain + ((u64)j * 2 + 1) * T + tid, acc);
    }
    fe inv;
    fe_inv(inv, acc);
    for (u32 j = pi - 1; j >= 1; j--) {
        fe hx, hy, d, s, t, lam, x, nhx;
        fe_load2(hx, helper + (u64)(j - 1) * 4 + 0, helper + (u64)(j - 1) * 4 + 1);
        fe_load2(hy, helper + (u64)(j - 1) * 4 + 2, helper + (u64)(j - 1) * 4 + 3);
        const bool dbl = fe_eq(hx, Sx);
        fe_sub(d, hx, Sx);
        if (__builtin_expect(dbl, 0)) fe_add(d, Sy, Sy);
        if (j > 1) {
            fe c;
            fe_load2(c, chain + ((u64)(j - 1) * 2 + 0) * T + tid, chain + ((u64)(j - 1) * 2 + 1) * T + tid);
            fe_mul(s, inv, c);
            fe_mul(inv, inv, d);
        } else {
            s = inv;
        }
        fe_sub(t, hy, Sy);
        if (__builtin_expect(dbl, 0)) { fe x2; fe_sqr(x2, Sx); fe_add(t, x2, x2); fe_add(t, t, x2); }
        fe_mul(lam, t, s);
        fe_neg(nhx, hx);
        x_from_lambda(x, lam, nSx, nhx);
        const u64 idx = (u64)j * T + tid;
        if (idx < count) keys[idx] = ((u64)x.v[1] << 32) | x.v[0];
    }
}

// sort key = bucket << 32 | hash ; value = position (197:2561, 2583, 1221)
__global__ void sortkeys_kernel(const u64 *__restrict__ keys, u64 *__restrict__ sk, u32 *__restrict__ pos, u64 n, u64 first_pos, u32 mask)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = keys[i];
        sk[i] = ((u64)((u32)k & mask) << 32) | (k >> 32);
        pos[i] = (u32)(first_pos + i);
    }
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

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return bsgs_big_malloc(&p, n ? n : 1); }      // parked scratch pieces are handed back on demand
    template <class T> T *as() { return (T *)p; }
};
}  // namespace

// core: images are written to DEVICE buffers gpu_img / cpu_img (either may be NULL)
static int build_to_device(bsgs_dev *d, uint64_t w, uint32_t htsz, u32 *gpu_img, u32 *cpu_img)
{
    const uint64_t ht_items = 1ull << htsz;
    // geometry of one generation chunk: T threads x pi points
    const uint32_t T = 1u << 16, pi = 512;
    const uint64_t chunk = (uint64_t)T * pi;
    DevBuf keys, sk, sk2, pos, pos2, chainb, helperb, basesb, tmp, offs;
    HIPCHK(keys.alloc(w * 8));
    HIPCHK(chainb.alloc((uint64_t)T * pi * 32));
    // helper j*(T*G), j = 1..pi-1  (host EC library; a few hundred points)
    const hs::Affine TG = hs::point_mul(hs::G, hs::fe_from_u64(T));
    {
        std::vector<hs::Affine> helper = hs::multiples(TG, pi - 1);
        std::vector<uint8_t> hb((size_t)(pi - 1) * 64);
        for (size_t i = 0; i + 1 < pi; i++) hs::affine_to_le(helper[i], &hb[i * 64], &hb[i * 64 + 32]);
        HIPCHK(helperb.alloc(hb.size()));
        HIPCHK(hipMemcpy(helperb.p, hb.data(), hb.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(basesb.alloc((size_t)T * 64));
    std::vector<uint8_t> bb((size_t)T * 64);
    for (uint64_t first = 1; first <= w; first += chunk) {
        const uint64_t count = std::min<uint64_t>(chunk, w - first + 1);
        // bases (first + tid)*G for tid < T: consecutive multiples from a scalar-multiplied start
        {
            const hs::Affine start = hs::point_mul(hs::G, hs::fe_from_u64(first));
            std::vector<hs::Jac> j(T);
            hs::Jac cur = hs::to_jac(start);
            for (uint32_t i = 0; i < T; i++) { j[i] = cur; cur = hs::jac_add_affine(cur, hs::G); }
            std::vector<hs::Affine> a = hs::batch_to_affine(j);
            for (uint32_t i = 0; i < T; i++) hs::affine_to_le(a[i], &bb[(size_t)i * 64], &bb[(size_t)i * 64 + 32]);
        }
        HIPCHK(hipMemcpyAsync(basesb.p, bb.data(), bb.size(), hipMemcpyHostToDevice, d->stream));
        hipLaunchKernelGGL(baby_keys_kernel, dim3(T / 256), dim3(256), 0, d->stream, helperb.as<const u32x4>(), basesb.as<const u32x4>(),
                           keys.as<u64>() + (first - 1), chainb.as<u32x4>(), T, pi, count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(d->stream));      // bb is reused by the next chunk
    }
    (void)hipFree(chainb.p); chainb.p = nullptr;
    // sort by (bucket, hash), positions ride along
    HIPCHK(sk.alloc(w * 8)); HIPCHK(sk2.alloc(w * 8)); HIPCHK(pos.alloc(w * 4)); HIPCHK(pos2.alloc(w * 4));
    const int gblocks = (int)std::min<uint64_t>((w + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(sortkeys_kernel, dim3(gblocks), dim3(256), 0, d->stream, keys.as<const u64>(), sk.as<u64>(), pos.as<u32>(), w, 0ull,
                       (u32)(ht_items - 1));
    HIPCHK(hipGetLastError());
    size_t tmp_bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, sk.as<u64>(), sk2.as<u64>(), pos.as<u32>(), pos2.as<u32>(), (size_t)w, 0u, 32u + htsz, d->stream));
    HIPCHK(tmp.alloc(tmp_bytes));
    HIPCHK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, sk.as<u64>(), sk2.as<u64>(), pos.as<u32>(), pos2.as<u32>(), (size_t)w, 0u, 32u + htsz, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    (void)hipFree(keys.p); keys.p = nullptr;
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

static int ext_check(bsgs_dev *d, uint64_t w, uint32_t htsz, uint32_t layout)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!w || w > (1ull << 36) || htsz < 1 || htsz > 31) return fail(BSGS_ERR_ARG, "need 0 < w <= 2^36 and 1 <= htsz <= 31");
    if (layout != BSGS_TABLE_LINES64_LIST && layout != BSGS_TABLE_LINES128_LIST) return fail(BSGS_ERR_ARG, "layout must be BSGS_TABLE_LINES64_LIST or BSGS_TABLE_LINES128_LIST");
    return BSGS_OK;
}

// upper estimate of the entries that will not fit their line (Poisson loads), with slack
static uint64_t ext_list_capacity(uint64_t w, uint32_t htsz, uint32_t layout)
{
    const unsigned cap_line = layout == BSGS_TABLE_LINES128_LIST ? 30 : 14;      // the last word of a full line is the bound of its overflow entries (ext_scatter_kernel)
    const double buckets = (double)(1ull << htsz);
    return std::min<uint64_t>(w, (uint64_t)(1.25 * expected_overflow_entries((double)w / buckets, cap_line, buckets)) + (1u << 20));
}

extern "C" int bsgs_ext_overflow_capacity(uint64_t w, uint32_t htsz, uint32_t layout, uint64_t *cap)
{
    if (!cap || !w || htsz < 1 || htsz > 31 || (layout != BSGS_TABLE_LINES64_LIST && layout != BSGS_TABLE_LINES128_LIST)) return fail(BSGS_ERR_ARG, "bad args");
    *cap = bsgs_ovf_slots(ext_list_capacity(w, htsz, layout));
    return BSGS_OK;
}

// the builder proper: lines = 2^htsz lines, ovf_table = ovf_slots u64 (both device memory)
static int ext_build_into(bsgs_dev *d, uint64_t w, uint32_t htsz, int lplog, u32x4 *lines, u64 *ovf_table, uint64_t ovf_slots, uint64_t *ovf_n, uint64_t *overflow_buckets)
{
    const uint64_t ovf_cap = ovf_slots / 2;                           // the hash set takes at most that many keys
    DevBuf listb;
    HIPCHK(listb.alloc(ovf_cap * 8));
    u64 *ovf = listb.as<u64>();
    const uint64_t ht_items = 1ull << htsz, line_bytes = 64ull << (lplog - 2);
    // one generation chunk = T threads x pi points; large chunks keep the host-side base points (T per chunk) off the clock
    const uint32_t T = 1u << 16, pi = w > (1ull << 28) ? 4096 : 512;
    const uint64_t chunk = (uint64_t)T * pi;
    size_t fr = 0, tot = 0;
    HIPCHK(bsgs_mem_available(&fr, &tot));
    const uint64_t need = std::min(chunk, w) * 8 + (uint64_t)T * pi * 32 + (64ull << 20);     // keys, chain
    if (need > fr) return fail(BSGS_ERR_NOMEM, "extended table build needs %.1f GiB of scratch, %.1f GiB free", need / 1073741824.0, fr / 1073741824.0);
    DevBuf keys, chainb, helperb, basesb, cnt;
    HIPCHK(hipMemsetAsync(lines, 0, ht_items * line_bytes, d->stream));
    HIPCHK(cnt.alloc(16));
    HIPCHK(hipMemsetAsync(cnt.p, 0, 16, d->stream));
    HIPCHK(keys.alloc(std::min(chunk, w) * 8));
    HIPCHK(chainb.alloc((uint64_t)T * pi * 32));
    const hs::Affine TG = hs::point_mul(hs::G, hs::fe_from_u64(T));
    {
        std::vector<hs::Affine> helper = hs::multiples(TG, pi - 1);
        std::vector<uint8_t> hb((size_t)(pi - 1) * 64);
        for (size_t i = 0; i + 1 < pi; i++) hs::affine_to_le(helper[i], &hb[i * 64], &hb[i * 64 + 32]);
        HIPCHK(helperb.alloc(hb.size()));
        HIPCHK(hipMemcpy(helperb.p, hb.data(), hb.size(), hipMemcpyHostToDevice));
    }
    HIPCHK(basesb.alloc((size_t)T * 64));
    std::vector<uint8_t> bb((size_t)T * 64);
    const u32 mask = (u32)(ht_items - 1);
    for (uint64_t first = 1; first <= w; first += chunk) {
        const uint64_t count = std::min<uint64_t>(chunk, w - first + 1);
        {
            const hs::Affine start = hs::point_mul(hs::G, hs::fe_from_u64(first));
            std::vector<hs::Jac> j(T);
            hs::Jac cur = hs::to_jac(start);
            for (uint32_t i = 0; i < T; i++) { j[i] = cur; cur = hs::jac_add_affine(cur, hs::G); }
            std::vector<hs::Affine> a = hs::batch_to_affine(j);
            HIPCHK(hipStreamSynchronize(d->stream));                 // bb is still the source of the previous chunk's copy
            for (uint32_t i = 0; i < T; i++) hs::affine_to_le(a[i], &bb[(size_t)i * 64], &bb[(size_t)i * 64 + 32]);
        }
        HIPCHK(hipMemcpyAsync(basesb.p, bb.data(), bb.size(), hipMemcpyHostToDevice, d->stream));
        hipLaunchKernelGGL(baby_keys_kernel, dim3(T / 256), dim3(256), 0, d->stream, helperb.as<const u32x4>(), basesb.as<const u32x4>(),
                           keys.as<u64>(), chainb.as<u32x4>(), T, pi, count);
        const int sblocks = (int)std::min<uint64_t>((count + 255) / 256, 1u << 16);
        if (lplog == 2) hipLaunchKernelGGL(ext_scatter_kernel<2>, dim3(sblocks), dim3(256), 0, d->stream, keys.as<const u64>(), count, mask, (u32 *)lines, ovf, ovf_cap, cnt.as<unsigned long long>());
        else            hipLaunchKernelGGL(ext_scatter_kernel<3>, dim3(sblocks), dim3(256), 0, d->stream, keys.as<const u64>(), count, mask, (u32 *)lines, ovf, ovf_cap, cnt.as<unsigned long long>());
        HIPCHK(hipGetLastError());
    }
    const int fblocks = (int)std::min<uint64_t>((ht_items + 255) / 256, 1u << 20);
    if (lplog == 2) hipLaunchKernelGGL(ext_finalize_kernel<2>, dim3(fblocks), dim3(256), 0, d->stream, (u32 *)lines, ht_items, cnt.as<unsigned long long>());
    else            hipLaunchKernelGGL(ext_finalize_kernel<3>, dim3(fblocks), dim3(256), 0, d->stream, (u32 *)lines, ht_items, cnt.as<unsigned long long>());
    HIPCHK(hipGetLastError());
    unsigned long long h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, cnt.p, 16, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (h[1] > ovf_cap) return fail(BSGS_ERR_NOMEM, "overflow list: %llu entries, capacity %llu", h[1], (unsigned long long)ovf_cap);
    (void)hipFree(chainb.p); chainb.p = nullptr;
    (void)hipFree(keys.p); keys.p = nullptr;
    if (h[1]) {
        // OVERFLOW BOUND (giant_kernel.hip.h): sort the overflow list by (bucket, hash), then per bucket keep the smallest hashes in the line
        // and put the smallest of the others into the line's last word
        DevBuf sorted, tmp;
        HIPCHK(sorted.alloc(h[1] * 8));
        size_t tmp_bytes = 0;
        HIPCHK(rocprim::radix_sort_keys(nullptr, tmp_bytes, ovf, sorted.as<u64>(), (size_t)h[1], 0u, 32u + htsz, d->stream));
        HIPCHK(tmp.alloc(tmp_bytes));
        HIPCHK(rocprim::radix_sort_keys(tmp.p, tmp_bytes, ovf, sorted.as<u64>(), (size_t)h[1], 0u, 32u + htsz, d->stream));
        HIPCHK(hipMemcpyAsync(ovf, sorted.p, h[1] * 8, hipMemcpyDeviceToDevice, d->stream));
        const int rblocks = (int)std::min<uint64_t>((h[1] + 255) / 256, 1u << 16);
        if (lplog == 2) hipLaunchKernelGGL(ext_refine_kernel<2>, dim3(rblocks), dim3(256), 0, d->stream, (u32 *)lines, ovf, (u64)h[1]);
        else            hipLaunchKernelGGL(ext_refine_kernel<3>, dim3(rblocks), dim3(256), 0, d->stream, (u32 *)lines, ovf, (u64)h[1]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(d->stream));
    }
    int rc = bsgs_ovf_fill(d, ovf, h[1], ovf_table, ovf_slots);
    if (rc) return rc;
    *ovf_n = ovf_slots; *overflow_buckets = h[0];
    return BSGS_OK;
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
    rc = bsgs_install_lines(d, (u32x4 *)lines_dev, layout == BSGS_TABLE_LINES128_LIST ? 3 : 2, (u64 *)ovf_dev, ovf_n, 1ull << htsz, w, overflow_buckets);
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
    const uint64_t ht_items = 1ull << htsz, line_bytes = layout == BSGS_TABLE_LINES128_LIST ? 128 : 64;
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
    const uint64_t ht_items = 1ull << htsz, line_bytes = 64ull << (lplog - 2);
    uint64_t ovf_cap = 0;
    rc = bsgs_ext_overflow_capacity(w, htsz, layout, &ovf_cap);
    if (rc) return rc;
    size_t fr = 0, tot = 0;
    HIPCHK(bsgs_mem_available(&fr, &tot));
    if (ht_items * line_bytes + ovf_cap * 8 > fr) return fail(BSGS_ERR_NOMEM, "extended table needs %.1f GiB, %.1f GiB free", (ht_items * line_bytes + ovf_cap * 8) / 1073741824.0, fr / 1073741824.0);
    u32x4 *lines = nullptr;
    u64 *ovf = nullptr;
    HIPCHK(bsgs_lines_malloc(d, (void **)&lines, ht_items * line_bytes));
    hipError_t e = bsgs_big_malloc((void **)&ovf, ovf_cap * 8);
    if (e != hipSuccess) { (void)hipFree(lines); return fail(BSGS_ERR_HIP, "hipMalloc overflow list: %s", hipGetErrorString(e)); }
    uint64_t n = 0, ob = 0;
    rc = ext_build_into(d, w, htsz, lplog, lines, ovf, ovf_cap, &n, &ob);
    if (rc) { (void)hipFree(lines); (void)hipFree(ovf); return rc; }
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
