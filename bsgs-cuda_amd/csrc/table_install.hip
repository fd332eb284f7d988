// table_install.hip -- the baby table on the device: upload / install (reference-format images and extended tables), replicas for several engines of one process,
// and the verification of what an engine holds (checksums, census, batched membership, sampled giants).  Part of libbsgs_hip.so (include/bsgs_hip.h).
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

// ---- baby table -----------------------------------------------------------------------------------------
static int validate_ext_table(bsgs_dev *d, const u32x4 *lines, int lplog, const u64 *ovf, uint64_t ovf_n, uint64_t ht_items);
// with_list: the entries that do not fit go to a sorted overflow list and the CSR image is dropped afterwards
static int build_lines(bsgs_dev *d, uint32_t layout, bool with_list)
{
    const int lplog = layout == BSGS_TABLE_LINES128 ? 3 : 2;
    d->lines_bytes = d->ht_items * (64ull << (lplog - 2));
    HIPCHK(bsgs_lines_malloc(d, (void **)&d->lines, d->lines_bytes));
    unsigned long long *cnt = nullptr, h[2] = {0, 0};
    HIPCHK(hipMalloc(&cnt, 16));
    const int blocks = (int)std::min<uint64_t>((d->ht_items + 255) / 256, 1u << 20);
    uint64_t cap = 0;
    u64 *list = nullptr;
    for (int pass = 0; pass < (with_list ? 2 : 1); pass++) {          // pass 0 of 2 only counts the overflow entries
        HIPCHK(hipMemsetAsync(cnt, 0, 16, d->stream));
        u64 *arg = with_list ? (pass ? list : (u64 *)cnt) : nullptr;   // any non-NULL pointer with capacity 0 in the counting pass
        if (lplog == 2) hipLaunchKernelGGL(lines_build_kernel<2>, dim3(blocks), dim3(256), 0, d->stream, d->csr, (u32 *)d->lines, d->ht_items, cnt, arg, cap);
        else            hipLaunchKernelGGL(lines_build_kernel<3>, dim3(blocks), dim3(256), 0, d->stream, d->csr, (u32 *)d->lines, d->ht_items, cnt, arg, cap);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h, cnt, 16, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (with_list && pass == 0) { cap = h[1]; HIPCHK(hipMalloc(&list, cap ? cap * 8 : 8)); }
    }
    (void)hipFree(cnt);
    d->overflow = h[0];
    d->layout = layout;
    if (with_list) {
        const uint64_t slots = bsgs_ovf_slots(cap);
        u64 *table = nullptr;
        if (hipMalloc(&table, slots * 8) != hipSuccess) { (void)hipFree(list); return fail(BSGS_ERR_NOMEM, "overflow set: %llu slots", (unsigned long long)slots); }
        int rc = bsgs_ovf_fill(d, list, cap, table, slots);
        (void)hipFree(list);
        if (rc) { (void)hipFree(table); return rc; }
        rc = validate_ext_table(d, d->lines, lplog, table, slots, d->ht_items);      // an image with unsorted buckets (not the reference's format) ends here
        if (rc) { (void)hipFree(table); (void)bsgs_big_free(d->lines); d->lines = nullptr; d->layout = 0; return rc; }
        d->ovf = table; d->ovf_n = slots; d->bound_copies = false;
        if (d->csr && d->csr_owned) (void)hipFree(d->csr);
        d->csr = nullptr;                                               // borrowed images stay with the caller
    }
    return BSGS_OK;
}

uint64_t bsgs_ovf_slots(uint64_t entries)
{
    uint64_t s = 2;
    while (s < 2 * entries) s <<= 1;
    return s;
}
int bsgs_ovf_fill(bsgs_dev *d, const u64 *list, uint64_t n, u64 *table, uint64_t slots)
{
    if (slots < 2 || (slots & (slots - 1)) || 2 * n > slots) return fail(BSGS_ERR_ARG, "overflow set: %llu keys do not fit %llu slots at load 1/2", (unsigned long long)n, (unsigned long long)slots);
    HIPCHK(hipMemsetAsync(table, 0xFF, slots * 8, d->stream));
    if (n) {
        const int blocks = (int)std::min<uint64_t>((n + 255) / 256, 1u << 16);
        hipLaunchKernelGGL(ovf_insert_kernel, dim3(blocks), dim3(256), 0, d->stream, list, n, table, slots - 1);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(d->stream));
    return BSGS_OK;
}

// the invariant the probe's overflow-bound shortcut rests on (giant_kernel.hip.h: ext_validate_*): checked for every lines + overflow-set table
static int validate_ext_table(bsgs_dev *d, const u32x4 *lines, int lplog, const u64 *ovf, uint64_t ovf_n, uint64_t ht_items)
{
    unsigned long long *bad = nullptr, h[2] = {0, 0};
    HIPCHK(hipMalloc(&bad, 16));
    hipError_t e = hipMemsetAsync(bad, 0, 16, d->stream);
    const int lb = (int)std::min<uint64_t>((ht_items + 255) / 256, 1u << 16), sb = (int)std::min<uint64_t>((ovf_n + 255) / 256, 1u << 16);
    if (lplog == 3) {
        hipLaunchKernelGGL(ext_validate_lines_kernel<3>, dim3(lb), dim3(256), 0, d->stream, (const u32 *)lines, ht_items, bad);
        hipLaunchKernelGGL(ext_validate_set_kernel<3>, dim3(sb), dim3(256), 0, d->stream, (const u32 *)lines, ht_items, ovf, ovf_n, bad);
    } else {
        hipLaunchKernelGGL(ext_validate_lines_kernel<2>, dim3(lb), dim3(256), 0, d->stream, (const u32 *)lines, ht_items, bad);
        hipLaunchKernelGGL(ext_validate_set_kernel<2>, dim3(sb), dim3(256), 0, d->stream, (const u32 *)lines, ht_items, ovf, ovf_n, bad);
    }
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, bad, 16, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(bad);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "table validation: %s", hipGetErrorString(e));
    if (h[0] || h[1])
        return fail(BSGS_ERR_ARG, "this lines + overflow-set table breaks the overflow bound (%llu over-full lines hold an entry above their last word, %llu keys of the set are "
                                  "below their line's last word, missing from its fingerprint or belong to no over-full line): a probe would miss entries.  Build it with bsgs_build_baby_table_ext*, or "
                                  "from an htGPU image whose buckets are sorted ascending", h[0], h[1]);
    return BSGS_OK;
}

int bsgs_install_lines(bsgs_dev *d, u32x4 *lines, int lplog, u64 *ovf, uint64_t ovf_n, uint64_t ht_items, uint64_t w,
                       uint64_t overflow_buckets)
{
    if (ht_items < 2 || ht_items >= (1ull << 32)) return fail(BSGS_ERR_ARG, "2 <= buckets < 2^32");
    if (ovf) { int rc = validate_ext_table(d, lines, lplog, ovf, ovf_n, ht_items); if (rc) return rc; }
    bsgs_free_table(d);
    d->lines = lines; d->lines_bytes = ht_items * (64ull << (lplog - 2));
    d->ovf = ovf; d->ovf_n = ovf_n; d->bound_copies = true;
    d->ht_items = ht_items; d->w = w; d->overflow = overflow_buckets;
    d->bucket_mul = (ht_items & (ht_items - 1)) ? (uint32_t)ht_items : 0u;      // any number of buckets: the multiplicative bucket function (giant_kernel.hip.h bucket_of)
    d->layout = lplog == 3 ? BSGS_TABLE_LINES128 : BSGS_TABLE_LINES64;
    return BSGS_OK;
}

static int finish_table(bsgs_dev *d, uint64_t ht_items, uint64_t w, uint32_t layout)
{
    d->ht_items = ht_items; d->w = w;
    if (layout == BSGS_TABLE_AUTO) {
        // mean bucket load decides the line size; fall back to CSR when the lines do not fit in free memory
        // up to 5 entries per bucket: 64-byte lines, the few over-full buckets through the resident CSR image; up to 9: still
        // 64-byte lines, but ~1 % of the probes then need the fallback, which has to be the hash set (1.5 reads, not a CSR
        // search): the CSR image is dropped; up to 20: 128-byte lines + hash set; beyond that the exact CSR probe
        const double load = (double)w / (double)ht_items;
        layout = load <= 5.0 ? BSGS_TABLE_LINES64 : load <= 9.0 ? BSGS_TABLE_LINES64_LIST : BSGS_TABLE_LINES128_LIST;
        size_t fr = 0, tot = 0;
        HIPCHK(bsgs_mem_available(&fr, &tot));     // parked scratch pieces are ours on demand: they must not push the table into the CSR layout
        const uint64_t need = ht_items * (layout == BSGS_TABLE_LINES128_LIST ? 128ull : 64ull);
        if (load > 20.0 || need + (1ull << 30) > fr) layout = BSGS_TABLE_CSR;
    }
    if (layout == BSGS_TABLE_CSR) { d->layout = BSGS_TABLE_CSR; d->lines_bytes = 0; d->overflow = 0; return BSGS_OK; }
    if (layout == BSGS_TABLE_LINES64_LIST) return build_lines(d, BSGS_TABLE_LINES64, true);
    if (layout == BSGS_TABLE_LINES128_LIST) return build_lines(d, BSGS_TABLE_LINES128, true);
    if (layout != BSGS_TABLE_LINES64 && layout != BSGS_TABLE_LINES128) return fail(BSGS_ERR_ARG, "unknown layout %u", layout);
    return build_lines(d, layout, false);
}

static int check_table_args(uint64_t ht_items, uint64_t w)
{
    if (!ht_items || (ht_items & (ht_items - 1))) return fail(BSGS_ERR_ARG, "ht_items must be a power of two");
    if (ht_items > (1ull << 32) || w >= (1ull << 32)) return fail(BSGS_ERR_ARG, "reference format limits: ht_items <= 2^32, w < 2^32");
    return BSGS_OK;
}

extern "C" int bsgs_upload_htgpu(bsgs_dev *d, const void *image, uint64_t ht_items, uint64_t w, uint32_t layout)
{
    if (!d || !image) return fail(BSGS_ERR_ARG, "null");
    int rc = check_table_args(ht_items, w);
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->id));
    bsgs_free_table(d);
    const uint64_t bytes = 4 * (ht_items + 1) + 4 * w;
    HIPCHK(bsgs_big_malloc(&d->csr, bytes));
    d->csr_owned = true;
    HIPCHK(hipMemcpy(d->csr, image, bytes, hipMemcpyHostToDevice));
    return finish_table(d, ht_items, w, layout);
}

extern "C" int bsgs_upload_htgpu_device(bsgs_dev *d, const void *dimage, uint64_t ht_items, uint64_t w, uint32_t layout)
{
    if (!d || !dimage) return fail(BSGS_ERR_ARG, "null");
    int rc = check_table_args(ht_items, w);
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->id));
    bsgs_free_table(d);
    d->csr = (u32 *)dimage;          // borrowed: the caller keeps the image alive (it is the overflow fallback)
    d->csr_owned = false;
    return finish_table(d, ht_items, w, layout);
}

extern "C" int bsgs_table_info(bsgs_dev *d, uint32_t *layout, uint64_t *device_bytes, uint64_t *overflow_buckets)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    if (layout) *layout = d->ovf ? d->layout + 2 : d->layout;        // 4 / 5: bucket lines + overflow list, no CSR image
    if (device_bytes) *device_bytes = d->ovf ? d->lines_bytes + 8 * d->ovf_n : 4 * (d->ht_items + 1) + 4 * d->w + d->lines_bytes;
    if (overflow_buckets) *overflow_buckets = d->overflow;
    return BSGS_OK;
}

// 1 = the engine owns its bucket lines / overflow set (built by it, or received into bsgs_alloc_table_ext_recv buffers); 0 = borrowed
extern "C" int bsgs_debug_table_owner(bsgs_dev *d, int *lines_owned)
{
    if (!d || !lines_owned) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    *lines_owned = d->lines ? (d->lines_owned ? 1 : 0) : (d->csr_owned ? 1 : 0);
    return BSGS_OK;
}


// ---- replicas for several GPUs of one process: the reference uploads G2 and htGPU to every GPU over PCIe (1_9_7File.pb:2337,
// 2350, 4769-4843).  Here devs[0] holds the giants and the table (file-backed or GPU-built) and every other engine gets its replica over
// xGMI: RCCL (one communicator per engine in this process, ncclBroadcast inside one group -- north_star's "RCCL over xGMI only to broadcast
// htGPU at startup") when the engines sit on distinct GPUs, else -- one GPU listed twice, no librccl -- direct peer copies, all destinations
// at once, each on its own stream (startup.hip: bsgs_fabric).  what: bit 0 = the giants, bit 1 = the table.
extern "C" int bsgs_broadcast_tables_ex(bsgs_dev *const *devs, int n, uint32_t transport, uint32_t what, uint32_t *transport_used, double *seconds)
{
    if (!devs || n < 1 || !devs[0]) return fail(BSGS_ERR_ARG, "null");
    if (!(what & 3u)) return fail(BSGS_ERR_ARG, "nothing to replicate (what = 1 giants | 2 table)");
    bsgs_dev *s = devs[0];
    if ((what & 1u) && !s->g2) return fail(BSGS_ERR_STATE, "devs[0] must hold the giants");
    if ((what & 2u) && !s->layout) return fail(BSGS_ERR_STATE, "devs[0] must hold the table");
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i < n; i++) {
        if (!devs[i]) return fail(BSGS_ERR_ARG, "null device %d", i);
        if (devs[i] == s) return fail(BSGS_ERR_ARG, "device %d is devs[0] itself", i);
    }
    bsgs_fabric *F = nullptr;
    int rc = bsgs_fabric_open(&F, devs, n, transport);
    if (rc) return rc;
    if (transport_used) *transport_used = bsgs_fabric_is_rccl(F) ? BSGS_TRANSPORT_RCCL : BSGS_TRANSPORT_PEER;
    // allocations first (every replica's buffers), then the transfers; a replica's table state (layout, sizes) is set only after every allocation and
    // copy for it succeeded: a failure half-way leaves a device WITHOUT a table (bsgs_enqueue then refuses), never one with a layout and null pointers
    std::vector<void *> g2(n, nullptr), csr(n, nullptr), lines(n, nullptr), ovf(n, nullptr);
    g2[0] = s->g2; csr[0] = s->csr; lines[0] = s->lines; ovf[0] = s->ovf;
    // (an engine whose prepare() has not run may still hold a BORROWED table -- a bsgs_share_tables twin, caller-owned buffers of bsgs_install_table_ext_device --
    // whose memory is not ours to free: only the engines prepare() allocated for are torn down)
    std::vector<bool> prepared(n, false);
    auto fail_all = [&](int code) {
        const std::string why = bsgs_last_error();
        for (int k = 1; k < n; k++) {
            if (!prepared[k]) continue;
            (void)hipSetDevice(devs[k]->id); (void)hipStreamSynchronize(devs[k]->stream);
            if (what & 2u) { devs[k]->lines_owned = true; devs[k]->csr_owned = true; bsgs_free_table(devs[k]); }
        }
        bsgs_fabric_close(F);
        return fail(code, "%s", why.c_str());
    };
    for (int i = 1; i < n; i++) {
        bsgs_dev *d = devs[i];
        auto prepare = [&]() -> int {
            HIPCHK(hipSetDevice(d->id));
            if (what & 1u) {
                int r = bsgs_set_geometry(d, s->t, s->b, s->p);
                if (r) return r;
                if (d->Ti != s->Ti || d->pi != s->pi) return fail(BSGS_ERR_STATE, "device %d chose another batching", i);
                g2[i] = d->g2;
            }
            if (what & 2u) {
                bsgs_free_table(d);                       // (a borrowed table is only forgotten here; from now on everything this engine holds was allocated below)
                prepared[i] = true;
                if (s->csr) { HIPCHK(bsgs_big_malloc(&d->csr, 4 * (s->ht_items + 1) + 4 * s->w)); d->csr_owned = true; csr[i] = d->csr; }
                if (s->lines) { HIPCHK(bsgs_lines_malloc(d, (void **)&d->lines, s->lines_bytes)); d->lines_owned = true; lines[i] = d->lines; }
                if (s->ovf) { HIPCHK(bsgs_big_malloc((void **)&d->ovf, s->ovf_n * 8)); d->ovf_n = s->ovf_n; ovf[i] = d->ovf; }
            }
            return BSGS_OK;
        };
        rc = prepare();
        if (rc) return fail_all(rc);
    }
    if (what & 1u) rc = bsgs_fabric_broadcast(F, g2.data(), s->maxnonce * 64, 0);
    if (rc == BSGS_OK && (what & 2u) && s->csr) rc = bsgs_fabric_broadcast(F, csr.data(), 4 * (s->ht_items + 1) + 4 * s->w, 0);
    if (rc == BSGS_OK && (what & 2u) && s->lines) rc = bsgs_fabric_broadcast(F, lines.data(), s->lines_bytes, 0);
    if (rc == BSGS_OK && (what & 2u) && s->ovf) rc = bsgs_fabric_broadcast(F, ovf.data(), s->ovf_n * 8, 0);
    if (rc) return fail_all(rc);
    if (what & 2u)
        for (int i = 1; i < n; i++) {
            bsgs_dev *d = devs[i];
            d->ht_items = s->ht_items; d->w = s->w; d->overflow = s->overflow; d->lines_bytes = s->lines_bytes; d->layout = s->layout; d->bucket_mul = s->bucket_mul; d->bound_copies = s->bound_copies;
        }
    bsgs_fabric_close(F);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return BSGS_OK;
}
// Two engines on one GPU probing ONE table (the host's lanes): the twin borrows the owner's table buffers and copies the giants.
extern "C" int bsgs_share_tables(bsgs_dev *s, bsgs_dev *d)
{
    if (!s || !d || s == d) return fail(BSGS_ERR_ARG, "two different engines");
    if (s->id != d->id) return fail(BSGS_ERR_ARG, "engines on GPU %d and GPU %d: a table is shared on ONE GPU only (replicas elsewhere: bsgs_broadcast_tables)", s->id, d->id);
    if (!s->g2 || !s->layout) return fail(BSGS_ERR_STATE, "the owner must hold the giants and the table");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued on the twin");
    HIPCHK(hipSetDevice(d->id));
    int r = bsgs_set_geometry(d, s->t, s->b, s->p);
    if (r) return r;
    if (d->Ti != s->Ti || d->pi != s->pi) return fail(BSGS_ERR_STATE, "the twin chose another batching");
    HIPCHK(hipStreamSynchronize(s->stream));                          // whatever built the owner's buffers is done
    HIPCHK(hipMemcpyAsync(d->g2, s->g2, s->maxnonce * 64, hipMemcpyDeviceToDevice, d->stream));
    bsgs_free_table(d);
    d->csr = s->csr; d->csr_owned = false;
    d->lines = s->lines; d->ovf = s->ovf; d->ovf_n = s->ovf_n; d->lines_owned = false;
    d->ht_items = s->ht_items; d->w = s->w; d->overflow = s->overflow; d->lines_bytes = s->lines_bytes; d->layout = s->layout; d->bucket_mul = s->bucket_mul; d->bound_copies = s->bound_copies;
    HIPCHK(hipStreamSynchronize(d->stream));
    return BSGS_OK;
}
extern "C" int bsgs_broadcast_tables(bsgs_dev *const *devs, int n)
{
    if (n == 1 && devs && devs[0]) return (devs[0]->g2 && devs[0]->layout) ? BSGS_OK : fail(BSGS_ERR_STATE, "devs[0] must hold the giants and the table");
    return bsgs_broadcast_tables_ex(devs, n, BSGS_TRANSPORT_AUTO, 3u, nullptr, nullptr);
}

// ---- replica verification -----------------------------------------------------------------------------------------------------
// The reference gives every GPU its own upload from host memory (1_9_7File.pb:2337, 2350, 4769-4843); here replicas come from a
// device-to-device copy (bsgs_broadcast_tables) or an RCCL broadcast (pybsgs.dist), and a replica that differs in one byte would lose keys
// silently.  So every holder reduces what it holds to 64-bit checksums ON THE DEVICE (one pass at streaming rate: 16 GiB of lines in ~5 ms)
// and the hosts compare them across engines / ranks (bsgs_mi355x -verifyreplicas, bench.py `table_checksum_equal`).
//   position-dependent: sum over 64-bit words v at index i of mix(v + i * golden)   -- bucket lines, CSR image, giants
//   position-independent: sum of mix(key) over the occupied slots                   -- the overflow hash set (slot order depends on insertion order)
__device__ __forceinline__ u64 ck_mix(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
template <bool POSITIONAL>
static __global__ void __launch_bounds__(256) checksum_kernel(const u64 *__restrict__ v, u64 n, const u32 *__restrict__ tail, unsigned long long *out)
{
    u64 acc = 0;
    if (tail && blockIdx.x == 0 && threadIdx.x == 0) acc = ck_mix((u64)*tail + n * 0x9E3779B97F4A7C15ULL);      // a buffer of 8n + 4 bytes: its last 32-bit word
    const u64 stride = (u64)gridDim.x * blockDim.x * 2;
    for (u64 i = (blockIdx.x * (u64)blockDim.x + threadIdx.x) * 2; i < n; i += stride) {
        if (i + 1 < n) {
            const ulonglong2 w = *(const ulonglong2 *)(v + i);          // 16 bytes per lane: one contiguous KiB per wave instruction
            if (POSITIONAL) acc += ck_mix(w.x + i * 0x9E3779B97F4A7C15ULL) + ck_mix(w.y + (i + 1) * 0x9E3779B97F4A7C15ULL);
            else acc += (w.x != BSGS_OVF_EMPTY ? ck_mix(w.x) : 0) + (w.y != BSGS_OVF_EMPTY ? ck_mix(w.y) : 0);
        } else {
            const u64 w = v[i];
            if (POSITIONAL) acc += ck_mix(w + i * 0x9E3779B97F4A7C15ULL);
            else acc += w != BSGS_OVF_EMPTY ? ck_mix(w) : 0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)acc);
}
static int checksum_of(bsgs_dev *d, const void *buf, uint64_t bytes, bool positional, unsigned long long *slot)
{
    if (!buf || bytes < 8) return BSGS_OK;
    const u64 n = bytes / 8;
    const u32 *tail = (bytes & 4) ? (const u32 *)buf + 2 * n : nullptr;   // the CSR image is (2^htsz + 1 + w) 32-bit words: possibly an odd number
    const int blocks = (int)std::min<uint64_t>((n / 2 + 255) / 256, (uint64_t)d->prop.multiProcessorCount * 16);
    if (positional) hipLaunchKernelGGL(checksum_kernel<true>, dim3(blocks), dim3(256), 0, d->stream, (const u64 *)buf, n, tail, slot);
    else            hipLaunchKernelGGL(checksum_kernel<false>, dim3(blocks), dim3(256), 0, d->stream, (const u64 *)buf, n, tail, slot);
    HIPCHK(hipGetLastError());
    return BSGS_OK;
}
extern "C" int bsgs_table_checksum(bsgs_dev *d, uint64_t sums[4])
{
    if (!d || !sums) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout && !d->g2) return fail(BSGS_ERR_STATE, "nothing on the device");
    HIPCHK(hipSetDevice(d->id));
    unsigned long long *acc = nullptr;
    HIPCHK(hipMalloc(&acc, 32));
    int rc = BSGS_OK;
    hipError_t e = hipMemsetAsync(acc, 0, 32, d->stream);
    if (e == hipSuccess && d->layout) {
        if (rc == BSGS_OK) rc = checksum_of(d, d->lines, d->lines ? d->lines_bytes : 0, true, acc + 0);
        if (rc == BSGS_OK) rc = checksum_of(d, d->ovf, d->ovf_n * 8, false, acc + 1);
        if (rc == BSGS_OK) rc = checksum_of(d, d->csr, d->csr ? 4 * (d->ht_items + 1) + 4 * d->w : 0, true, acc + 2);
    }
    if (e == hipSuccess && rc == BSGS_OK && d->g2) rc = checksum_of(d, d->g2, d->maxnonce * 64, true, acc + 3);
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, acc, 32, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(acc);
    if (rc) return rc;
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "table checksum: %s", hipGetErrorString(e));
    for (int k = 0; k < 4; k++) sums[k] = h[k];
    return BSGS_OK;
}
// ---- structural verification of the installed table (the reference's checkHT / checkHTpack, 1_9_7File.pb:3599-3627, 3101-3134, 2797-2805) ----------------
// out[0] entries held by bucket lines (or by the CSR image: over-full buckets of BSGS_TABLE_LINES64 / 128, everything of BSGS_TABLE_CSR), [1] over-full lines,
// [2] keys in the overflow set, [3] duplicates (a line's last word that is also a key of the set), [4] malformed lines / buckets, [5] lines / buckets not ascending,
// [6] w as installed, [7] out[0] + out[2] - out[3]: must equal [6].  One streaming pass (128 GiB of lines: 40 ms).
extern "C" int bsgs_table_census(bsgs_dev *d, uint64_t out[8])
{
    if (!d || !out) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    HIPCHK(hipSetDevice(d->id));
    unsigned long long *c = nullptr, h[6] = {0, 0, 0, 0, 0, 0};
    HIPCHK(hipMalloc(&c, sizeof h));
    hipError_t e = hipMemsetAsync(c, 0, sizeof h, d->stream);
    const int blocks = (int)std::min<uint64_t>((d->ht_items + 255) / 256, (uint64_t)d->prop.multiProcessorCount * 32);
    if (d->lines) {
        if (d->layout == BSGS_TABLE_LINES128) hipLaunchKernelGGL(table_census_kernel<3>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->lines, d->ht_items, (const u32 *)d->csr, (const u64 *)d->ovf, d->ovf_n, c, d->bound_copies);
        else                                  hipLaunchKernelGGL(table_census_kernel<2>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->lines, d->ht_items, (const u32 *)d->csr, (const u64 *)d->ovf, d->ovf_n, c, d->bound_copies);
        if (d->ovf) hipLaunchKernelGGL(set_census_kernel, dim3((int)std::min<uint64_t>((d->ovf_n + 255) / 256, (uint64_t)d->prop.multiProcessorCount * 32)), dim3(256), 0, d->stream, (const u64 *)d->ovf, d->ovf_n, c);
    } else hipLaunchKernelGGL(csr_census_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32 *)d->csr, d->ht_items, c);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, c, sizeof h, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(c);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "table census: %s", hipGetErrorString(e));
    for (int k = 0; k < 6; k++) out[k] = h[k];
    out[6] = d->w; out[7] = h[0] + h[2] - h[3];
    return BSGS_OK;
}

// Batched membership through the shipped probe: found[i] = 1 when the tile kernel would report a hit for the 64-bit key keys64[i] (low 64 bits of an x coordinate:
// bucket from the low word, hash = the high word).  Host buffers; n keys, n bytes.
extern "C" int bsgs_table_lookup(bsgs_dev *d, const uint64_t *keys64, uint64_t n, uint8_t *found)
{
    if (!d || !keys64 || !found) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    if (!n) return BSGS_OK;
    if (n > (1ull << 31)) return fail(BSGS_ERR_ARG, "at most 2^31 keys per call");
    HIPCHK(hipSetDevice(d->id));
    u64 *dk = nullptr; unsigned char *df = nullptr;
    HIPCHK(hipMalloc(&dk, n * 8));
    if (hipMalloc(&df, n) != hipSuccess) { (void)hipFree(dk); return fail(BSGS_ERR_NOMEM, "lookup buffers"); }
    TileArgs A = {};
    A.csr = d->csr; A.lines = d->lines; A.ovf = d->ovf; A.ovf_n = d->ovf_n; A.ht_items = d->ht_items; A.ht_mask = (u32)(d->ht_items - 1); A.bucket_mul = d->bucket_mul;
    hipError_t e = hipMemcpyAsync(dk, keys64, n * 8, hipMemcpyHostToDevice, d->stream);
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    if (d->layout == BSGS_TABLE_LINES64 && d->bucket_mul) hipLaunchKernelGGL(table_lookup_kernel<4>, grid, block, 4096, d->stream, A, (const u64 *)dk, (u64)n, df);
    else if (d->layout == BSGS_TABLE_LINES64)  hipLaunchKernelGGL(table_lookup_kernel<2>, grid, block, 4096, d->stream, A, (const u64 *)dk, (u64)n, df);
    else if (d->layout == BSGS_TABLE_LINES128) hipLaunchKernelGGL(table_lookup_kernel<3>, grid, block, 8192, d->stream, A, (const u64 *)dk, (u64)n, df);
    else                                       hipLaunchKernelGGL(table_lookup_kernel<0>, grid, block, 0, d->stream, A, (const u64 *)dk, (u64)n, df);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(found, df, n, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(dk); (void)hipFree(df);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "table lookup: %s", hipGetErrorString(e));
    return BSGS_OK;
}

// sampled giants as plain points (64 bytes x_le || y_le each): what the host compares with (i + 1) * ADDPUBG before it searches, like the reference's
// checkGiantArr (1_9_7File.pb:1524-1559, called :1941).  idx: n giant numbers in [0, t*b*p); host buffers.
extern "C" int bsgs_sample_g2(bsgs_dev *d, const uint64_t *idx, uint32_t n, uint8_t *out_xy_le)
{
    if (!d || !idx || !out_xy_le) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device");
    if (!n) return BSGS_OK;
    for (uint32_t k = 0; k < n; k++) if (idx[k] >= d->maxnonce) return fail(BSGS_ERR_ARG, "giant %llu of %llu", (unsigned long long)idx[k], (unsigned long long)d->maxnonce);
    HIPCHK(hipSetDevice(d->id));
    u64 *di = nullptr; fe *dout = nullptr;
    HIPCHK(hipMalloc(&di, (size_t)n * 8));
    if (hipMalloc(&dout, (size_t)n * 64) != hipSuccess) { (void)hipFree(di); return fail(BSGS_ERR_NOMEM, "sample buffers"); }
    hipError_t e = hipMemcpyAsync(di, idx, (size_t)n * 8, hipMemcpyHostToDevice, d->stream);
    hipLaunchKernelGGL(g2_sample_kernel, dim3((n + 63) / 64), dim3(64), 0, d->stream, (const u32x4 *)d->g2, d->Ti, d->pi, (const u64 *)di, n, dout);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out_xy_le, dout, (size_t)n * 64, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(di); (void)hipFree(dout);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "sample giants: %s", hipGetErrorString(e));
    return BSGS_OK;
}
