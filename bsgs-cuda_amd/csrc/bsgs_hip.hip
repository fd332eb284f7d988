// bsgs_hip.hip -- C-ABI (include/bsgs_hip.h) of the MI355X giant-step engine: native API.
// Build: see ../Makefile (hipcc --offload-arch=gfx950 -shared -fPIC).  No CPU fallback exists: every
// entry point fails with BSGS_ERR_HIP when no gfx950 device / runtime is available.
#include "bsgs_internal.h"
#include "support_kernels.hip.h"
#include "host_secp.h"

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;
int bsgs_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

extern "C" const char *bsgs_last_error(void) { return g_err.c_str(); }
extern "C" const char *bsgs_version(void) { return "bsgs-hip 0.4 (gfx950)"; }
// The -D switches this library was built with, space separated; "" for the shipped build.  "WRONG-RESULTS:" prefixes the *_CEILING
// switches (timing experiments whose hit lists are wrong by construction: they compile only with -DBSGS_EXPERIMENT).
extern "C" const char *bsgs_build_info(void)
{
    static const std::string info = [] {
        std::string s;
        auto add = [&](const char *n) { if (!s.empty()) s += ' '; s += n; };
#define SW(name) add(#name)
#define WRONG(name) add("WRONG-RESULTS:" #name)
#ifdef BSGS_EXPERIMENT
        SW(BSGS_EXPERIMENT);
#endif
#ifdef BSGS_NO_OVF_CEILING
        WRONG(BSGS_NO_OVF_CEILING);
#endif
#ifdef BSGS_QUAD_CEILING
        WRONG(BSGS_QUAD_CEILING);
#endif
#ifdef BSGS_NOCHAIN_CEILING
        WRONG(BSGS_NOCHAIN_CEILING);
#endif
#ifdef BSGS_NOCHAIN_STORE_CEILING
        WRONG(BSGS_NOCHAIN_STORE_CEILING);
#endif
#ifdef BSGS_NOCHAIN_LOAD_CEILING
        WRONG(BSGS_NOCHAIN_LOAD_CEILING);
#endif
#ifdef BSGS_OCT_CEILING
        WRONG(BSGS_OCT_CEILING);
#endif
#ifdef BSGS_G2_DUP_CEILING
        WRONG(BSGS_G2_DUP_CEILING);
#endif
#ifdef BSGS_G2_CACHED_CEILING
        WRONG(BSGS_G2_CACHED_CEILING);
#endif
#ifdef BSGS_FULL_X
        SW(BSGS_FULL_X);
#endif
#ifdef BSGS_INV_PER_WAVE
        SW(BSGS_INV_PER_WAVE);
#endif
#ifdef FE_FOLD_EXACT_ONLY
        SW(FE_FOLD_EXACT_ONLY);
#endif
#ifdef FE_FOLD_C
        SW(FE_FOLD_C);
#endif
#ifdef FE_FOLD_COLUMNS
        SW(FE_FOLD_COLUMNS);
#endif
#ifdef FE_SQR_VIA_MUL
        SW(FE_SQR_VIA_MUL);
#endif
        if (BSGS_PAIR2_WAVES != 4) add("BSGS_PAIR2_WAVES=" BSGS_STR(BSGS_PAIR2_WAVES));
        if (BSGS_PAIR2_WAVES128 != 3) add("BSGS_PAIR2_WAVES128=" BSGS_STR(BSGS_PAIR2_WAVES128));
        if (BSGS_TILE_CHUNK != 64u) add("BSGS_TILE_CHUNK=" BSGS_STR(BSGS_TILE_CHUNK));
        if (BSGS_NT_CHAIN != 1) add("BSGS_NT_CHAIN=" BSGS_STR(BSGS_NT_CHAIN));
        if (BSGS_NT_LINES != 0) add("BSGS_NT_LINES=" BSGS_STR(BSGS_NT_LINES));
        if (BSGS_PROBE_CPOL != 2) add("BSGS_PROBE_CPOL=" BSGS_STR(BSGS_PROBE_CPOL));
#ifdef BSGS_SLICE_GATE
        add("BSGS_SLICE_GATE=" BSGS_STR(BSGS_SLICE_GATE));
#endif
#undef SW
#undef WRONG
        return s;
    }();
    return info.c_str();
}

static void release_pending(bsgs_dev *d);

extern "C" int bsgs_dev_count(int *n)
{
    if (!n) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipGetDeviceCount(n));
    return BSGS_OK;
}

extern "C" int bsgs_dev_open(int device_id, bsgs_dev **out)
{
    if (!out) return fail(BSGS_ERR_ARG, "null");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(BSGS_ERR_ARG, "device %d of %d", device_id, n);
    HIPCHK(hipSetDevice(device_id));
    bsgs_dev *d = new bsgs_dev();
    d->id = device_id;
    if (const char *v = getenv("BSGS_KERNEL_VARIANT")) {                          // A-B and tests only; the variants are bit-identical
        const int k = atoi(v);
        if (k != 13 && k != 10 && k != 0) { delete d; return fail(BSGS_ERR_ARG, "BSGS_KERNEL_VARIANT=%d: this library has 13 (default), 10 and 0", k); }
        d->variant = k;
    }
    HIPCHK(hipGetDeviceProperties(&d->prop, device_id));
    HIPCHK(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&d->ev0));
    HIPCHK(hipEventCreate(&d->ev1));
    if (const char *v = getenv("BSGS_DEBUG_PHASES")) d->debug_flags = (unsigned)atoi(v);     // timing experiments only
    if (const char *v = getenv("BSGS_NARROW_LAUNCHES")) d->narrow_env_off = atoi(v) == 0; // A-B only: 0 = every launch with the default batching
    HIPCHK(hipMalloc(&d->hitbuf, bsgs_hitbuf_bytes(d)));
    HIPCHK(hipHostMalloc(&d->hit_host, bsgs_hitbuf_bytes(d), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(d->hitbuf, 0, 64, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *out = d;
    return BSGS_OK;
}

static void free_table(bsgs_dev *d)
{
    if (d->csr && d->csr_owned) (void)hipFree(d->csr);
    if (d->lines && d->lines_owned) (void)bsgs_big_free(d->lines);
    if (d->ovf && d->lines_owned) (void)hipFree(d->ovf);
    d->csr = nullptr; d->lines = nullptr; d->ovf = nullptr; d->ovf_n = 0; d->layout = 0; d->lines_owned = true; d->auto_tpl = 0; d->bucket_mul = 0;
    d->narrow_off = false;                  // memory was short for a narrow copy of the giants ONCE: another table, another try
}
void bsgs_free_table(bsgs_dev *d) { free_table(d); }
void bsgs_free_recv(bsgs_dev *d)
{
    if (d->recv_lines) (void)bsgs_big_free(d->recv_lines);
    if (d->recv_ovf) (void)hipFree(d->recv_ovf);
    d->recv_lines = d->recv_ovf = nullptr;
}
static void free_narrow(bsgs_dev *d)
{
    for (auto &b : d->narrow) if (b.g2) (void)hipFree(b.g2);
    d->narrow.clear();
}
static void free_g2(bsgs_dev *d)
{
    if (d->g2) (void)hipFree(d->g2);
    free_narrow(d);
    if (d->chain) (void)hipFree(d->chain);
    free_chain_pieces(d);
    if (d->quirk_list) (void)hipFree(d->quirk_list);
    d->quirk_list = nullptr; d->quirk_host.clear(); d->quirk_ready = false;
    d->g2 = nullptr; d->chain = nullptr; d->chain_bytes = 0;
}

extern "C" int bsgs_dev_close(bsgs_dev *d)
{
    if (!d) return BSGS_OK;
    (void)hipSetDevice(d->id);
    (void)hipStreamSynchronize(d->stream);
    release_pending(d);
    free_table(d); free_g2(d); bsgs_free_recv(d);
    release_grader(d);
    free_reserve(d);
    park_release(d->id);
    if (d->hitbuf) (void)hipFree(d->hitbuf);
    if (d->hit_host) (void)hipHostFree(d->hit_host);
    if (d->cen_dev) (void)hipFree(d->cen_dev);
    if (d->cen_pin) (void)hipHostFree(d->cen_pin);
    if (d->walk_table) (void)hipFree(d->walk_table);
    if (d->digest) (void)hipFree(d->digest);
    (void)hipEventDestroy(d->ev0); (void)hipEventDestroy(d->ev1);
    (void)hipStreamDestroy(d->stream);
    delete d;
    return BSGS_OK;
}

extern "C" int bsgs_dev_name(bsgs_dev *d, char *buf, int len)
{
    if (!d || !buf || len <= 0) return fail(BSGS_ERR_ARG, "bad args");
    snprintf(buf, (size_t)len, "%s (%s)", d->prop.name, d->prop.gcnArchName);
    return BSGS_OK;
}
extern "C" int bsgs_dev_meminfo(bsgs_dev *d, uint64_t *fr, uint64_t *tot)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    size_t f = 0, t = 0;
    HIPCHK(bsgs_mem_available(&f, &t));       // what this process has parked on the device is handed back on demand: it counts as free
    if (fr) *fr = f;
    if (tot) *tot = t;
    return BSGS_OK;
}
extern "C" int bsgs_dev_cu_count(bsgs_dev *d, int *cus)
{
    if (!d || !cus) return fail(BSGS_ERR_ARG, "null");
    *cus = d->prop.multiProcessorCount;
    return BSGS_OK;
}
extern "C" int bsgs_dev_stream(bsgs_dev *d, void **s)
{
    if (!d || !s) return fail(BSGS_ERR_ARG, "null");
    *s = (void *)d->stream;
    return BSGS_OK;
}
extern "C" int bsgs_steps_per_tile(bsgs_dev *d, uint64_t *steps)
{
    if (!d || !steps) return fail(BSGS_ERR_ARG, "null");
    *steps = 2 * d->maxnonce;
    return BSGS_OK;
}

extern "C" int bsgs_set_tiles_per_launch(bsgs_dev *d, uint32_t n)
{
    if (!d || n > BSGS_TILES_PER_LAUNCH_MAX) return fail(BSGS_ERR_ARG, "tiles per launch must be 0 (auto) or 1..%d", BSGS_TILES_PER_LAUNCH_MAX);
    d->tiles_per_launch = n;
    return BSGS_OK;
}
extern "C" int bsgs_tiles_per_launch(bsgs_dev *d, uint32_t *n)
{
    if (!d || !n) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device: the launch shape follows the geometry");
    *n = bsgs_auto_tiles_per_launch(d);
    return BSGS_OK;
}
extern "C" int bsgs_engine_geometry(bsgs_dev *d, uint32_t *threads, uint32_t *giants_per_thread)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device");
    if (threads) *threads = d->Ti;
    if (giants_per_thread) *giants_per_thread = d->pi;
    return BSGS_OK;
}
extern "C" int bsgs_set_flags(bsgs_dev *d, uint32_t flags)
{
    if (!d || (flags & ~BSGS_FLAG_REFERENCE_QUIRKS)) return fail(BSGS_ERR_ARG, "unknown flag bits %#x", flags);
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    d->flags = flags;
    return BSGS_OK;
}
// the tile-kernel instantiation the most recent launch used, as rocprofv3 names it (the parity tests assert they ran the SHIPPED one)
extern "C" int bsgs_debug_last_batching(bsgs_dev *d, uint32_t *threads, uint32_t *giants_per_thread)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (threads) *threads = d->last_Ti;
    if (giants_per_thread) *giants_per_thread = d->last_pi;
    return BSGS_OK;
}

extern "C" int bsgs_debug_last_kernel(bsgs_dev *d, char *buf, int len)
{
    if (!d || !buf || len <= 0) return fail(BSGS_ERR_ARG, "bad args");
    snprintf(buf, (size_t)len, "%s", d->last_kernel);
    return BSGS_OK;
}
extern "C" int bsgs_launch_count(bsgs_dev *d, uint64_t *launches)
{
    if (!d || !launches) return fail(BSGS_ERR_ARG, "null");
    *launches = d->launches;
    return BSGS_OK;
}

// ---- giants ------------------------------------------------------------------------------------------
int bsgs_set_geometry(bsgs_dev *d, uint32_t t, uint32_t b, uint32_t p)
{
    if (!t || !b || !p) return fail(BSGS_ERR_ARG, "t,b,p must be non-zero");
    const uint64_t T = (uint64_t)t * b, maxnonce = T * p;
    if (T >= (1ull << 31) || maxnonce >= (1ull << 32)) return fail(BSGS_ERR_ARG, "t*b*p must be < 2^32 (hit index is u32)");
    HIPCHK(hipSetDevice(d->id));
    free_g2(d);
    d->t = t; d->b = b; d->p = p; d->T = T; d->maxnonce = maxnonce; d->auto_tpl = 0; d->narrow_off = false;
    // Internal batching: one Fermat inversion (270 multiplications) is shared by pi giants of a thread, so a
    // longer batch is cheaper per giant step; the thread count lost that way is won back by putting more tiles
    // in one launch.  Grow pi up to ~2048 while Ti stays a multiple of 256 threads and >= 2048.
    uint32_t m = 1;
    if (const char *e = getenv("BSGS_BATCH_MULT")) m = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 1;     // tuning / A-B only
    else while ((uint64_t)p * m * 2 <= 1024 && T % (2ull * m) == 0 && T / (2ull * m) >= 2048 && (T / (2ull * m)) % 256 == 0) m *= 2;
    if (T % m) m = 1;
    d->Ti = (uint32_t)(T / m); d->pi = p * m;
    HIPCHK(bsgs_big_malloc(&d->g2, maxnonce * 64));
    return BSGS_OK;
}

// threads per workgroup of the tile kernel: four waves (10 KiB of LDS per wave with 64-byte lines: four blocks fill a CU; the 128-byte-line kernel, compiled for three waves
// per SIMD, runs three such blocks per CU).  BSGS_LINES128_BLOCK=128 gives the 128-byte-line kernel two-wave blocks (A-B: six blocks per CU instead of three).
static unsigned tile_block(const bsgs_dev *d)
{
    static const unsigned b128 = getenv("BSGS_LINES128_BLOCK") ? (unsigned)atoi(getenv("BSGS_LINES128_BLOCK")) : 256u;
    return d->layout == BSGS_TABLE_LINES128 && (b128 == 128u || b128 == 256u) ? b128 : d->block_size;
}
// giants per stored running product of the tile kernel a launch with batch length `pi` takes: 4 / 2 = the chained kernel (giant_pair2_kernel,
// QUAD or not), 1 = the per-giant kernel (CSR layout, odd batch lengths, BSGS_KERNEL_VARIANT=0)
uint32_t bsgs_chain_group(const bsgs_dev *d, uint32_t pi)
{
    if (d->variant == 0 || !bsgs_lines_layout(d) || (pi & 1u)) return 1;
    return (d->variant == 13 && (pi & 3u) == 0) ? 4 : 2;
}

// the prefix-product scratch per tile in flight: 32 bytes per giant for the per-giant kernel, 16 / 8 for the chained kernel (one stored
// product per two / four giants); `full` = the caller is a generator kernel that needs the per-giant chain of one tile
static int ensure_chain(bsgs_dev *d, uint64_t tiles, bool full = false)
{
    const uint32_t group = full ? 1 : bsgs_chain_group(d, d->pi);
    const bool chained = group > 1;
    // the tiles' scratch areas are 2^28 bytes apart at the usual geometry; tiles of a launch touch the same offsets at about the
    // same time, so a pad breaks the power-of-two stride between them (BSGS_CHAIN_PAD bytes, chained kernel only: an experiment that changed nothing)
    static const uint64_t pad_env = getenv("BSGS_CHAIN_PAD") ? strtoull(getenv("BSGS_CHAIN_PAD"), nullptr, 10) : 0;
    d->chain_pad = chained ? (uint32_t)(pad_env / 16) : 0;
    // the chained kernel's scratch is [tile][block][group][2][block size]: whole blocks (the tail block is padded)
    const uint64_t threads_padded = ((uint64_t)d->Ti + tile_block(d) - 1) / tile_block(d) * tile_block(d);
    const uint64_t per_tile = (chained ? threads_padded * d->pi * (32 / group) : d->maxnonce * 32) + (uint64_t)d->chain_pad * 16;
    const uint64_t bytes = per_tile * tiles;
    static const bool pieces_on = !(getenv("BSGS_CHAIN_PIECES") && atoi(getenv("BSGS_CHAIN_PIECES")) == 0);
    if (pieces_on && chained && bytes >= (8ull << 30) && per_tile <= (4ull << 30)) {
        uint32_t lg = 0;
        while ((per_tile << (lg + 1)) <= (4ull << 30)) lg++;                       // pieces of 2^lg tiles, at most 4 GiB
        const uint64_t piece_bytes = per_tile << lg, npieces = (tiles + (1ull << lg) - 1) >> lg;
        if (npieces <= BSGS_CHAIN_PIECES_MAX) {
            if (!d->chain_pieces.empty() && d->chain_piece_bytes == piece_bytes && d->chain_piece_log == lg && d->chain_pieces.size() >= npieces) return BSGS_OK;
            HIPCHK(hipStreamSynchronize(d->stream));
            free_chain_pieces(d);
            if (d->chain) { (void)hipFree(d->chain); d->chain = nullptr; }
            d->chain_bytes = 0;
            bool got = alloc_graded_pieces(d, npieces, piece_bytes);
            if (!got && !d->narrow.empty()) {                                       // the narrow copies of the giants are a convenience: they go first
                free_narrow(d);
                got = alloc_graded_pieces(d, npieces, piece_bytes);
            }
            if (!got) {
                size_t fr = 0, tot = 0;
                (void)bsgs_mem_available(&fr, &tot);
                return fail(BSGS_ERR_NOMEM, "chain scratch: %llu pieces of %.1f GiB for %llu tiles in flight, %.1f of %.1f GiB free", (unsigned long long)npieces,
                            piece_bytes / 1073741824.0, (unsigned long long)tiles, fr / 1073741824.0, tot / 1073741824.0);
            }
            d->chain_piece_bytes = piece_bytes; d->chain_piece_log = lg;
            d->chain_bytes = piece_bytes * npieces;
            return BSGS_OK;
        }
    }
    if (!d->chain_pieces.empty()) {                                                 // back to one buffer (a generator kernel, the per-giant kernel)
        HIPCHK(hipStreamSynchronize(d->stream));
        free_chain_pieces(d);
        d->chain_bytes = 0;
    }
    if (d->chain && d->chain_bytes >= bytes) return BSGS_OK;
    if (d->chain) { HIPCHK(hipStreamSynchronize(d->stream)); (void)hipFree(d->chain); d->chain = nullptr; d->chain_bytes = 0; }
    hipError_t e = bsgs_big_malloc(&d->chain, bytes);
    if (e != hipSuccess && !d->narrow.empty()) { (void)hipGetLastError(); (void)hipStreamSynchronize(d->stream); free_narrow(d); e = bsgs_big_malloc(&d->chain, bytes); }
    if (e != hipSuccess) {
        size_t fr = 0, tot = 0;
        (void)hipMemGetInfo(&fr, &tot);
        d->chain = nullptr;
        return fail(BSGS_ERR_NOMEM, "chain scratch: %.1f GiB for %llu tiles in flight, %.1f of %.1f GiB free", bytes / 1073741824.0,
                    (unsigned long long)tiles, fr / 1073741824.0, tot / 1073741824.0);
    }
    d->chain_bytes = bytes;
    return BSGS_OK;
}
uint32_t bsgs_auto_tiles_per_launch(const bsgs_dev *d)
{
    if (d->tiles_per_launch) return d->tiles_per_launch;
    if (d->auto_tpl) return d->auto_tpl;
    // One launch = three rounds of resident blocks at least (4 waves per SIMD fill the chip): 48 tiles of 16384 engine threads.
    // A launch boundary (ramp: every resident block in the streaming-bound prefix phase; tail) costs ~1.5 ms, i.e. 3.5 % at 48
    // tiles (44.7 ms); the centres live in device memory, so nothing but the chain scratch (8 bytes x giants per tile in flight; 16 with the pair chain)
    // limits a launch: measured 36.0 / 37.3 / 37.5 G giant-steps/s at 48 / 96 / 192 tiles (profiles/r02a_ab_tiles_per_launch.log).
    // Take 4x the fill-the-chip figure when that scratch fits in a third of the free memory, else 2x, else 1x.
    const uint64_t want = (uint64_t)d->prop.multiProcessorCount * 3072;
    uint64_t n = std::min<uint64_t>(std::max<uint64_t>((want + d->Ti - 1) / d->Ti, 1), BSGS_TILES_PER_LAUNCH);
    size_t fr = 0, tot = 0;
    const uint64_t per_giant = 32 / bsgs_chain_group(d, d->pi);                          // as ensure_chain sizes the scratch
    if (bsgs_mem_available(&fr, &tot) == hipSuccess) {
        fr += d->chain_bytes + d->group0_reserve.size() * d->group0_piece_bytes;     // what is already ours (scratch, reserve) counts as available
        // ... and tiles smaller than the usual 2^24 giants (the reference's README runs -t 256 -b 88 -p 130: 2.9 M) get more of them, so that a launch is the
        // same WORK -- 192 x 2^24 giants -- whatever the geometry: 192 tiles of 2.9 M giants are 30 ms launches and 37.3 G (profiles/r05h_*), the boundary
        // costs what it costs.  Up to BSGS_TILES_PER_LAUNCH_MAX tiles, memory permitting as before.
        const uint64_t work = std::min<uint64_t>(std::max<uint64_t>((192ull << 24) / std::max<uint64_t>(d->maxnonce, 1), n * 4), BSGS_TILES_PER_LAUNCH_MAX);
        // (a launch of few tiles may add ONE narrow copy of the giants, 64 bytes per giant: pick_batching; it is built only while twice that is free)
        for (uint64_t cand = work; cand > n; cand = (cand + 1) / 2)
            if (cand * d->maxnonce * per_giant <= fr / 3) { n = cand; break; }
    }
    const_cast<bsgs_dev *>(d)->auto_tpl = (uint32_t)n;           // decided once per geometry / table (reset by set_geometry, free_table)
    return (uint32_t)n;
}

extern "C" int bsgs_upload_g2_device(bsgs_dev *d, const void *dimage, uint32_t t, uint32_t b, uint32_t p)
{
    if (!d || !dimage) return fail(BSGS_ERR_ARG, "null");
    int rc = bsgs_set_geometry(d, t, b, p);
    if (rc) return rc;
    const int blocks = (int)std::min<uint64_t>((d->maxnonce + 255) / 256, 65535);
    hipLaunchKernelGGL(g2_relayout_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32 *)dimage, d->g2, (u32)d->T, p, d->Ti, d->pi);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(d->stream));
    return BSGS_OK;
}

extern "C" int bsgs_upload_g2(bsgs_dev *d, const void *image, uint32_t t, uint32_t b, uint32_t p)
{
    if (!d || !image) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    const uint64_t bytes = (uint64_t)t * b * p * 64;
    void *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, bytes));
    hipError_t e = hipMemcpy(tmp, image, bytes, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? bsgs_upload_g2_device(d, tmp, t, b, p) : fail(BSGS_ERR_HIP, "memcpy: %s", hipGetErrorString(e));
    (void)hipFree(tmp);
    return rc;
}

extern "C" int bsgs_download_g2(bsgs_dev *d, void *image_out, size_t bytes)
{
    if (!d || !image_out) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device");
    if (bytes != d->maxnonce * 64) return fail(BSGS_ERR_ARG, "image must be %llu bytes", (unsigned long long)(d->maxnonce * 64));
    HIPCHK(hipSetDevice(d->id));
    void *tmp = nullptr;
    HIPCHK(hipMalloc(&tmp, bytes));
    const int blocks = (int)std::min<uint64_t>((d->maxnonce + 255) / 256, 65535);
    hipLaunchKernelGGL(g2_to_image_kernel, dim3(blocks), dim3(256), 0, d->stream, d->g2, (u32 *)tmp, (u32)d->T, d->p, d->Ti, d->pi);
    hipError_t e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipMemcpy(image_out, tmp, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "download: %s", hipGetErrorString(e));
    return BSGS_OK;
}

extern "C" int bsgs_generate_g2(bsgs_dev *d, const uint8_t a_xy_le[64], uint32_t t, uint32_t b, uint32_t p)
{
    if (!d || !a_xy_le) return fail(BSGS_ERR_ARG, "null");
    int rc = bsgs_set_geometry(d, t, b, p);
    if (rc) return rc;
    // host: helper j*A (j = 1..p-1) and bases (tid*p+1)*A, via the host EC library (host_secp.h)
    hs::Affine A = hs::affine_from_le(a_xy_le, a_xy_le + 32);
    const uint32_t pi = d->pi, Ti = d->Ti;
    std::vector<hs::Affine> helper = hs::multiples(A, pi > 1 ? pi - 1 : 1);            // 1A..(pi-1)A
    std::vector<hs::Affine> bases = hs::strided_multiples(A, 1, pi, Ti);                // (1 + tid*pi) A
    std::vector<uint8_t> hbuf((size_t)std::max<uint32_t>(pi - 1, 1) * 64), bbuf((size_t)Ti * 64);
    for (size_t i = 0; i + 1 < pi; i++) hs::affine_to_le(helper[i], &hbuf[i * 64], &hbuf[i * 64 + 32]);
    for (size_t i = 0; i < Ti; i++) hs::affine_to_le(bases[i], &bbuf[i * 64], &bbuf[i * 64 + 32]);
    void *dh = nullptr, *db = nullptr;
    HIPCHK(hipMalloc(&dh, hbuf.size()));
    HIPCHK(hipMalloc(&db, bbuf.size()));
    HIPCHK(hipMemcpy(dh, hbuf.data(), hbuf.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db, bbuf.data(), bbuf.size(), hipMemcpyHostToDevice));
    const int blocks = (int)((Ti + 255) / 256);
    int rcc = ensure_chain(d, 1, true);
    if (rcc) { (void)hipFree(dh); (void)hipFree(db); return rcc; }
    hipLaunchKernelGGL(g2_generate_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)dh, (const u32x4 *)db,
                       d->g2, d->chain, Ti, pi);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(dh); (void)hipFree(db);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "g2_generate: %s", hipGetErrorString(e));
    return BSGS_OK;
}


// ---- tiles ------------------------------------------------------------------------------------------------
// grow-only device + pinned buffers for the centres of the queued tiles
static int ensure_centres(bsgs_dev *d, uint64_t tiles)
{
    if (tiles <= d->cen_cap) return BSGS_OK;
    uint64_t cap = d->cen_cap ? d->cen_cap : 4096;
    while (cap < tiles) cap *= 2;
    fe *nd = nullptr; uint8_t *np = nullptr;
    HIPCHK(hipMalloc(&nd, cap * 64));
    if (hipHostMalloc(&np, cap * 64, hipHostMallocDefault) != hipSuccess) { (void)hipFree(nd); return fail(BSGS_ERR_NOMEM, "pinned centre staging"); }
    if (d->cen_dev) d->pending_dev.push_back(d->cen_dev);       // launches in flight still read the old buffer
    if (d->cen_pin) d->pending_pinned.push_back(d->cen_pin);
    d->cen_dev = nd; d->cen_pin = np; d->cen_cap = cap;
    return BSGS_OK;
}

// ---- small launches ------------------------------------------------------------------------------------
// The default batching trades threads for batch length (set_geometry: 16384 threads x 1024 giants at -t 256 -b 256 -p 256) and wins the threads back
// with many tiles per launch.  A launch of ONE tile -- the reference's own launch pattern (1_9_7File.pb:2442-2459), what route A does whenever the
// centres cannot be predicted, what bsgs_step is -- then occupies 64 blocks of a 256-CU GPU: 6.5 G giant steps/s.  Any factorisation Ti' x pi' of
// maxnonce numbers the giants the same way (i = thread * pi' + slot), so such a launch takes a second copy of the giants laid out for shorter
// batches and more threads: the longest batch (>= 128 giants: below that the Fermat inversion -- 279 multiplications on the critical path of every block --
// costs more than the occupancy brings: 262144 x 64 runs one tile at 24.6 G, 131072 x 128 at 26.2 G) that still gives the launch four blocks per CU.  profiles/r04n_one_tile_launch_batching.log: 1 tile 6.5 -> 25.8 G, 4 tiles 25.3 -> 33.2 G.
// the rule itself (no device needed: tests/test_abi.py drives it through bsgs_debug_narrow_batching)
static uint32_t narrow_pi(uint64_t maxnonce, uint32_t pi, uint32_t ntiles, uint32_t cus, uint32_t block)
{
    const uint64_t target = (uint64_t)cus * 1024;                                                // four blocks of 256 threads per CU
    if (!pi || !block || (pi & 3u) || maxnonce % pi) return pi;
    while ((uint64_t)ntiles * (maxnonce / pi) < target && pi / 2 >= 128 && ((pi / 2) & 3u) == 0 &&
           maxnonce / (pi / 2) < (1ull << 31) && (maxnonce / (pi / 2)) % block == 0) pi /= 2;
    return pi;
}
extern "C" int bsgs_debug_narrow_batching(uint64_t giants_per_tile, uint32_t default_giants_per_thread, uint32_t ntiles, uint32_t cus, uint32_t block,
                                          uint32_t *giants_per_thread)
{
    if (!giants_per_thread) return fail(BSGS_ERR_ARG, "null");
    *giants_per_thread = narrow_pi(giants_per_tile, default_giants_per_thread, ntiles, cus, block);
    return BSGS_OK;
}
// At most ONE narrow copy is resident (64 bytes per giant): a launch size that wants another batching replaces it (the compat layer's 4 / 8 / 16-tile
// ramp and a ragged last launch would otherwise collect three), and bsgs_prepare builds the one-tile copy at start-up so that a job's clock never
// contains the allocation and the re-batching pass.
static const bsgs_dev::Batching *pick_batching(bsgs_dev *d, uint32_t ntiles)
{
    if (d->narrow_off || d->narrow_env_off || d->digest || d->debug_flags || d->phase_probe) return nullptr;
    if (bsgs_chain_group(d, d->pi) != 4) return nullptr;
    if ((d->flags & BSGS_FLAG_REFERENCE_QUIRKS) && !d->quirk_host.empty()) return nullptr;        // the quirk list is indexed by the default batching
    const uint32_t pi = narrow_pi(d->maxnonce, d->pi, ntiles, (uint32_t)d->prop.multiProcessorCount, d->block_size);      // (the rule is stated for 256-thread blocks; a two-wave block divides whatever it allows)
    if (pi == d->pi) return nullptr;
    for (const auto &b : d->narrow) if (b.pi == pi) return &b;
    // build it: 64 bytes per giant once more.  Not at the expense of anything else: only while twice that much (and 2 GiB) is free
    const uint64_t bytes = d->maxnonce * 64;
    bsgs_dev::Batching nb;
    nb.pi = pi; nb.Ti = (uint32_t)(d->maxnonce / pi);
    if (!d->narrow.empty()) {
        // re-use the resident copy's memory.  Launches in flight may still read it -- they were queued on this very stream, and so is the re-batching pass below:
        // stream order makes it wait for them, the host does not have to (a synchronisation here turned an asynchronous bsgs_enqueue into a blocking call: ADVICE r04)
        nb.g2 = d->narrow[0].g2;
        d->narrow.clear();
    } else {
        size_t fr = 0, tot = 0;
        if (bsgs_mem_available(&fr, &tot) != hipSuccess || fr < 2 * bytes + (2ull << 30)) { d->narrow_off = true; return nullptr; }
        if (bsgs_big_malloc(&nb.g2, bytes) != hipSuccess) { (void)hipGetLastError(); d->narrow_off = true; return nullptr; }     // like d->g2
    }
    const int blocks = (int)std::min<uint64_t>((d->maxnonce + 255) / 256, 65535);
    hipLaunchKernelGGL(g2_rebatch_kernel, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->g2, d->Ti, d->pi, nb.g2, nb.Ti, nb.pi, d->maxnonce);
    if (hipGetLastError() != hipSuccess) { (void)hipFree(nb.g2); d->narrow_off = true; return nullptr; }
    d->narrow.push_back(nb);
    return &d->narrow.back();
}

static int quirk_prepare(bsgs_dev *d);
static int launch_tiles(bsgs_dev *d, const fe *centres_dev, uint32_t ntiles, uint32_t seq)
{
    TileArgs A;
    hipStream_t st = d->stream;
    const bsgs_dev::Batching *nb = pick_batching(d, ntiles);
    const uint32_t Ti = nb ? nb->Ti : d->Ti, pi = nb ? nb->pi : d->pi;
    d->last_Ti = Ti; d->last_pi = pi;
    A.g2 = nb ? nb->g2 : d->g2; A.chain = d->chain; A.csr = d->csr; A.lines = d->lines; A.ovf = d->ovf; A.ovf_n = d->ovf_n; A.hitbuf = d->hitbuf;
    A.ht_items = d->ht_items; A.ht_mask = (u32)(d->ht_items - 1); A.pparam = pi; A.T = Ti;
    A.max_hits = d->max_hits; A.tile_seq = seq; A.ntiles = ntiles;
    A.debug_flags = d->debug_flags; A.bucket_mul = d->bucket_mul;
    A.centres_dev = centres_dev;
    A.digest = d->digest ? d->digest + (uint64_t)seq * Ti * 2 : nullptr;
    A.chain_pad = d->chain_pad; A.chain_mode = 0;
    for (int k = 0; k < BSGS_CHAIN_PIECES_MAX; k++) A.chain_piece[k] = nullptr;
    if (!d->chain_pieces.empty()) {                        // chained kernel (ensure_chain)
        A.chain = nullptr; A.chain_mode = d->chain_piece_log + 1;
        for (size_t k = 0; k < d->chain_pieces.size(); k++) A.chain_piece[k] = d->chain_pieces[k];
    }
    const unsigned bs = tile_block(d);
    if ((d->flags & BSGS_FLAG_REFERENCE_QUIRKS) && !d->quirk_host.empty()) {
        // the reference's own P - G arithmetic for the listed giants; bsgs_collect drops the hot loop's records for them
        const uint32_t nfix = (uint32_t)d->quirk_host.size() * ntiles;
        hipLaunchKernelGGL(quirk_fix_kernel, dim3((nfix + 63) / 64), dim3(64), 0, st, A, d->layout == BSGS_TABLE_LINES128 ? 3 : 2,
                           (const u32 *)d->quirk_list, (u32)d->quirk_host.size());
        HIPCHK(hipGetLastError());
    }
    const dim3 grid((unsigned)(((Ti + bs - 1) / bs) * ntiles)), block(bs);
    const uint32_t group = bsgs_chain_group(d, pi);
    // chained kernel, per wave: two probe slots (QUAD: one probe slot + the two 2 KiB temporaries) + the 2 KiB S stash: 4 blocks fill the 160 KiB of a CU exactly
    const bool l128 = d->layout == BSGS_TABLE_LINES128;
    const size_t slot = l128 ? 8192 : 4096;
    const size_t lds = group > 1 ? (size_t)(bs / 64) * ((group == 4 ? (l128 ? slot : slot + 4096) : 2 * slot) + 2048) : 0;       // giant_pair2_kernel: REGION + 2 KiB of stash per wave
    const bool dbg = d->debug_flags != 0 || d->phase_probe;
    if (d->layout == BSGS_TABLE_LINES64 && d->bucket_mul) HIPCHK(bsgs_launch_tile_lines64_any(A, grid, block, lds, st, group, dbg, &d->last_kernel));
    else if (d->layout == BSGS_TABLE_LINES64) HIPCHK(bsgs_launch_tile_lines64(A, grid, block, lds, st, group, dbg, &d->last_kernel));
    else if (l128)                       HIPCHK(bsgs_launch_tile_lines128(A, grid, block, lds, st, group, dbg, &d->last_kernel));
    else { hipLaunchKernelGGL((giant_tile_kernel<0>), grid, block, 0, st, A); d->last_kernel = "giant_tile_kernel<0>"; }
    HIPCHK(hipGetLastError());
    return BSGS_OK;
}

static void release_pending(bsgs_dev *d)
{
    for (void *p : d->pending_dev) (void)hipFree(p);
    for (void *p : d->pending_pinned) (void)hipHostFree(p);
    d->pending_dev.clear(); d->pending_pinned.clear();
}

// giants whose Gy trips the reference's NEGMODP (quirk mode): listed once per G2 upload
static int quirk_prepare(bsgs_dev *d)
{
    if (d->quirk_ready) return BSGS_OK;
    const uint32_t cap = 1u << 16;                           // 2.3e-7 of < 2^32 giants: about 1000 at most
    u32 *cnt = nullptr;
    if (!d->quirk_list) HIPCHK(hipMalloc(&d->quirk_list, (size_t)cap * 4));
    HIPCHK(hipMalloc(&cnt, 4));
    HIPCHK(hipMemsetAsync(cnt, 0, 4, d->stream));
    const int blocks = (int)std::min<uint64_t>((d->maxnonce + 255) / 256, 65535);
    hipLaunchKernelGGL(quirk_scan_kernel, dim3(blocks), dim3(256), 0, d->stream, d->g2, d->Ti, d->pi, d->quirk_list, cap, cnt);
    uint32_t n = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    (void)hipFree(cnt);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "quirk scan: %s", hipGetErrorString(e));
    if (n > cap) return fail(BSGS_ERR_NOMEM, "quirk list: %u giants, room for %u", n, cap);
    d->quirk_host.resize(n);
    if (n) HIPCHK(hipMemcpy(d->quirk_host.data(), d->quirk_list, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::sort(d->quirk_host.begin(), d->quirk_host.end());
    if (n) HIPCHK(hipMemcpy(d->quirk_list, d->quirk_host.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    d->quirk_ready = true;
    return BSGS_OK;
}

// how many giants of the resident G2 trip the reference's NEGMODP (the list reference-quirk mode re-computes after every launch)
extern "C" int bsgs_quirk_count(bsgs_dev *d, uint32_t *listed)
{
    if (!d || !listed) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    HIPCHK(hipSetDevice(d->id));
    int rc = quirk_prepare(d);
    if (rc) return rc;
    *listed = (uint32_t)d->quirk_host.size();
    return BSGS_OK;
}

// queue `ntiles` tiles whose centres are already in d->cen_dev[2*queued ...]
static int enqueue_common(bsgs_dev *d, uint32_t ntiles)
{
    if (d->flags & BSGS_FLAG_REFERENCE_QUIRKS) { int rq = quirk_prepare(d); if (rq) return rq; }
    const uint32_t tpl = bsgs_auto_tiles_per_launch(d);
    int rcc = ensure_chain(d, tpl);
    if (rcc) return rcc;
    if (!d->timing_open) {
        HIPCHK(hipEventRecord(d->ev0, d->stream));
        d->timing_open = true;
    }
    for (uint32_t k = 0; k < ntiles; k += tpl) {
        const uint32_t n = std::min<uint32_t>(tpl, ntiles - k);
        int rc = launch_tiles(d, d->cen_dev + 2 * (uint64_t)(d->queued + k), n, d->queued + k);
        if (rc) return rc;
        d->launches++;
    }
    d->queued += ntiles;
    return BSGS_OK;
}

// Allocate what the first launch would allocate -- the chain scratch for the launch size in effect, placed by grade -- NOW, as part of the
// start-up (the reference allocates its one buffer before the search loop too: 1_9_7File.pb:2251), so that a job's clock measures the search.
extern "C" int bsgs_prepare(bsgs_dev *d)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    HIPCHK(hipSetDevice(d->id));
    int rc = ensure_chain(d, bsgs_auto_tiles_per_launch(d));
    if (rc) return rc;
    (void)pick_batching(d, 1);               // the narrow copy of the giants a one-tile launch takes (bsgs_step, route A before its centres are predictable), memory permitting
    HIPCHK(hipStreamSynchronize(d->stream));
    return BSGS_OK;
}

extern "C" int bsgs_enqueue(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles)
{
    if (!d || !centres) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (!ntiles) return BSGS_OK;
    HIPCHK(hipSetDevice(d->id));
    int rc = ensure_centres(d, (uint64_t)d->queued + ntiles);
    if (rc) return rc;
    uint8_t *pin = d->cen_pin + (size_t)d->queued * 64;       // a fresh region per enqueue: nothing queued is overwritten
    memcpy(pin, centres, (size_t)ntiles * 64);
    HIPCHK(hipMemcpyAsync(d->cen_dev + 2 * (uint64_t)d->queued, pin, (size_t)ntiles * 64, hipMemcpyHostToDevice, d->stream));
    return enqueue_common(d, ntiles);
}

// ---- device-side tile walk: replaces GetJob's host point addition + the per-launch upload (1_9_7File.pb:2077-2092, 2435-2445) ----
extern "C" int bsgs_set_walk(bsgs_dev *d, const uint8_t p0_xy_le[64], const uint8_t stride_xy_le[64])
{
    if (!d || !p0_xy_le || !stride_xy_le) return fail(BSGS_ERR_ARG, "null");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    HIPCHK(hipSetDevice(d->id));
    hs::Affine D = hs::affine_from_le(stride_xy_le, stride_xy_le + 32), P0 = hs::affine_from_le(p0_xy_le, p0_xy_le + 32);
    if (!hs::on_curve(D) || !hs::on_curve(P0)) return fail(BSGS_ERR_ARG, "walk: P0 / stride is not a curve point");
    std::vector<uint8_t> tab(64 * 64);
    hs::Affine cur = D;
    for (int j = 0; j < 64; j++) {
        if (cur.inf) return fail(BSGS_ERR_ARG, "walk: 2^%d * stride is the point at infinity", j);
        hs::affine_to_le(cur, &tab[(size_t)j * 64], &tab[(size_t)j * 64 + 32]);
        cur = hs::point_add(cur, cur);
    }
    if (!d->walk_table) HIPCHK(hipMalloc(&d->walk_table, tab.size()));
    HIPCHK(hipMemcpy(d->walk_table, tab.data(), tab.size(), hipMemcpyHostToDevice));
    bsgs_le_to_fe(d->walk_p0x, p0_xy_le); bsgs_le_to_fe(d->walk_p0y, p0_xy_le + 32);
    d->walk_set = true;
    return BSGS_OK;
}

extern "C" int bsgs_enqueue_walk(bsgs_dev *d, uint64_t first_tile, uint32_t ntiles)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->walk_set) return fail(BSGS_ERR_STATE, "bsgs_set_walk first");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (!ntiles) return BSGS_OK;
    if (first_tile + ntiles < first_tile) return fail(BSGS_ERR_ARG, "tile index overflows 64 bits");
    HIPCHK(hipSetDevice(d->id));
    int rc = ensure_centres(d, (uint64_t)d->queued + ntiles);
    if (rc) return rc;
    hipLaunchKernelGGL(walk_centres_kernel, dim3((ntiles + 63) / 64), dim3(64), 0, d->stream, d->walk_p0x, d->walk_p0y,
                       (const fe *)d->walk_table, (u64)first_tile, (u32)ntiles, d->cen_dev + 2 * (uint64_t)d->queued, d->hitbuf + BSGS_HIT_WALK_STATUS);
    HIPCHK(hipGetLastError());
    return enqueue_common(d, ntiles);
}

extern "C" int bsgs_run_walk(bsgs_dev *d, uint64_t first_tile, uint32_t ntiles, bsgs_hit_ex *hits, uint32_t max_hits,
                             uint32_t *nhits, float *kernel_ms)
{
    int rc = bsgs_enqueue_walk(d, first_tile, ntiles);
    if (rc) return rc;
    return bsgs_collect(d, hits, max_hits, nhits, kernel_ms);
}

// the centres bsgs_enqueue_walk would use, for callers that need a tile's centre on the host (resolving a hit) and for tests
extern "C" int bsgs_walk_centres(bsgs_dev *d, uint64_t first_tile, uint32_t ntiles, uint8_t *centres_out)
{
    if (!d || !centres_out) return fail(BSGS_ERR_ARG, "null");
    if (!d->walk_set) return fail(BSGS_ERR_STATE, "bsgs_set_walk first");
    if (!ntiles) return BSGS_OK;
    HIPCHK(hipSetDevice(d->id));
    fe *tmp = nullptr; u32 *st = nullptr;
    HIPCHK(hipMalloc(&tmp, (size_t)ntiles * 64));
    if (hipMalloc(&st, 4) != hipSuccess) { (void)hipFree(tmp); return fail(BSGS_ERR_NOMEM, "status word"); }
    hipError_t e = hipMemsetAsync(st, 0, 4, d->stream);
    hipLaunchKernelGGL(walk_centres_kernel, dim3((ntiles + 63) / 64), dim3(64), 0, d->stream, d->walk_p0x, d->walk_p0y,
                       (const fe *)d->walk_table, (u64)first_tile, (u32)ntiles, tmp, st);
    if (e == hipSuccess) e = hipGetLastError();
    uint32_t bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, st, 4, hipMemcpyDeviceToHost, d->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d->stream);
    if (e == hipSuccess) e = hipMemcpy(centres_out, tmp, (size_t)ntiles * 64, hipMemcpyDeviceToHost);
    (void)hipFree(tmp); (void)hipFree(st);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "walk_centres: %s", hipGetErrorString(e));
    if (bad) return fail(BSGS_ERR_DEGENERATE, "%u tile centre(s) are the point at infinity", bad);
    return BSGS_OK;
}

extern "C" int bsgs_collect(bsgs_dev *d, bsgs_hit_ex *hits, uint32_t max_hits, uint32_t *nhits, float *kernel_ms)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    HIPCHK(hipSetDevice(d->id));
    if (d->timing_open) HIPCHK(hipEventRecord(d->ev1, d->stream));
    HIPCHK(hipMemcpyAsync(d->hit_host, d->hitbuf, 64, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    uint32_t n = d->hit_host[0];
    const uint32_t raw_n = n;
    const uint32_t walk_bad = d->hit_host[BSGS_HIT_WALK_STATUS];
    if (kernel_ms) {
        *kernel_ms = 0.f;
        if (d->timing_open) {
            HIPCHK(hipEventElapsedTime(kernel_ms, d->ev0, d->ev1));
        }
    }
    d->timing_open = false;
    d->queued = 0;
    release_pending(d);
    const uint32_t stored = std::min(n, d->max_hits);
    if (stored) {
        HIPCHK(hipMemcpyAsync(d->hit_host + BSGS_HIT_HEADER_WORDS, d->hitbuf + BSGS_HIT_HEADER_WORDS, (size_t)stored * 16,
                              hipMemcpyDeviceToHost, d->stream));
    }
    if (n || walk_bad) HIPCHK(hipMemsetAsync(d->hitbuf, 0, 64, d->stream));       // the host zeroes the counter after draining (1_9_7File.pb:2502-2503)
    HIPCHK(hipStreamSynchronize(d->stream));
    if (walk_bad) {
        if (nhits) *nhits = 0;
        return fail(BSGS_ERR_DEGENERATE, "%u tile centre(s) of the device walk are the point at infinity: dispense this batch with host centres", walk_bad);
    }
    const bsgs_hit_ex *rec = (const bsgs_hit_ex *)(d->hit_host + BSGS_HIT_HEADER_WORDS);
    std::vector<bsgs_hit_ex> v(rec, rec + stored);
    if ((d->flags & BSGS_FLAG_REFERENCE_QUIRKS) && !d->quirk_host.empty()) {
        // reference-quirk mode: for the listed giants the P - G record comes from quirk_fix_kernel (word 3 = 1), not from the hot loop
        size_t k = 0;
        uint32_t dropped = 0;
        for (size_t i = 0; i < v.size(); i++) {
            const bool listed = v[i].code == 2 && std::binary_search(d->quirk_host.begin(), d->quirk_host.end(), v[i].idx);
            if (listed && v[i].reserved == 0) { dropped++; continue; }
            v[k] = v[i]; v[k].reserved = 0; k++;
        }
        v.resize(k);
        n -= std::min(n, dropped);
    }
    if (nhits) *nhits = n;
    const uint32_t kept = (uint32_t)v.size();
    std::sort(v.begin(), v.end(), [](const bsgs_hit_ex &x, const bsgs_hit_ex &y) {
        if (x.tile != y.tile) return x.tile < y.tile;
        if (x.idx != y.idx) return x.idx < y.idx;
        return x.code < y.code;
    });
    const uint32_t ncopy = std::min<uint32_t>(kept, max_hits);
    if (hits && ncopy) memcpy(hits, v.data(), (size_t)ncopy * sizeof(bsgs_hit_ex));
    if (n > max_hits || raw_n > d->max_hits) return fail(BSGS_ERR_OVERFLOW, "%u hits, room for %u", raw_n, std::min(max_hits, d->max_hits));
    return BSGS_OK;
}

extern "C" int bsgs_run(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles, bsgs_hit_ex *hits, uint32_t max_hits,
                        uint32_t *nhits, float *kernel_ms)
{
    int rc = bsgs_enqueue(d, centres, ntiles);
    if (rc) return rc;
    return bsgs_collect(d, hits, max_hits, nhits, kernel_ms);
}

extern "C" int bsgs_step(bsgs_dev *d, const uint8_t px_le[32], const uint8_t py_le[32], bsgs_hit *hits, uint32_t max_hits,
                         uint32_t *nhits)
{
    if (!d || !px_le || !py_le) return fail(BSGS_ERR_ARG, "null");
    uint8_t c[64];
    memcpy(c, px_le, 32); memcpy(c + 32, py_le, 32);
    std::vector<bsgs_hit_ex> ex(max_hits ? max_hits : 1);
    uint32_t n = 0;
    int rc = bsgs_run(d, c, 1, ex.data(), max_hits, &n, nullptr);
    if (nhits) *nhits = n;
    if (rc && rc != BSGS_ERR_OVERFLOW) return rc;
    const uint32_t m = std::min(n, max_hits);
    for (uint32_t i = 0; i < m && hits; i++) { hits[i].code = ex[i].code; hits[i].idx = ex[i].idx; }
    return rc;
}
