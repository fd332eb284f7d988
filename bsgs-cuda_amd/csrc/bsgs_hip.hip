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

static size_t hitbuf_bytes(const bsgs_dev *d) { return 64 + (size_t)d->max_hits * 16; }

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
    HIPCHK(hipMalloc(&d->hitbuf, hitbuf_bytes(d)));
    HIPCHK(hipHostMalloc(&d->hit_host, hitbuf_bytes(d), hipHostMallocDefault));
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
static uint32_t auto_tiles_per_launch(const bsgs_dev *d);
extern "C" int bsgs_tiles_per_launch(bsgs_dev *d, uint32_t *n)
{
    if (!d || !n) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2) return fail(BSGS_ERR_STATE, "no giants on device: the launch shape follows the geometry");
    *n = auto_tiles_per_launch(d);
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
static int set_geometry(bsgs_dev *d, uint32_t t, uint32_t b, uint32_t p)
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

static bool lines_layout(const bsgs_dev *d) { return d->layout == BSGS_TABLE_LINES64 || d->layout == BSGS_TABLE_LINES128; }
// threads per workgroup of the tile kernel: four waves (10 KiB of LDS per wave with 64-byte lines: four blocks fill a CU; the 128-byte-line kernel, compiled for three waves
// per SIMD, runs three such blocks per CU).  BSGS_LINES128_BLOCK=128 gives the 128-byte-line kernel two-wave blocks (A-B: six blocks per CU instead of three).
static unsigned tile_block(const bsgs_dev *d)
{
    static const unsigned b128 = getenv("BSGS_LINES128_BLOCK") ? (unsigned)atoi(getenv("BSGS_LINES128_BLOCK")) : 256u;
    return d->layout == BSGS_TABLE_LINES128 && (b128 == 128u || b128 == 256u) ? b128 : d->block_size;
}
// giants per stored running product of the tile kernel a launch with batch length `pi` takes: 4 / 2 = the chained kernel (giant_pair2_kernel,
// QUAD or not), 1 = the per-giant kernel (CSR layout, odd batch lengths, BSGS_KERNEL_VARIANT=0)
static uint32_t chain_group(const bsgs_dev *d, uint32_t pi)
{
    if (d->variant == 0 || !lines_layout(d) || (pi & 1u)) return 1;
    return (d->variant == 13 && (pi & 3u) == 0) ? 4 : 2;
}

// the prefix-product scratch per tile in flight: 32 bytes per giant for the per-giant kernel, 16 / 8 for the chained kernel (one stored
// product per two / four giants); `full` = the caller is a generator kernel that needs the per-giant chain of one tile
static int ensure_chain(bsgs_dev *d, uint64_t tiles, bool full = false)
{
    const uint32_t group = full ? 1 : chain_group(d, d->pi);
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
static uint32_t auto_tiles_per_launch(const bsgs_dev *d)
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
    const uint64_t per_giant = 32 / chain_group(d, d->pi);                          // as ensure_chain sizes the scratch
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
    int rc = set_geometry(d, t, b, p);
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
    int rc = set_geometry(d, t, b, p);
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
        d->ovf = table; d->ovf_n = slots;
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
    free_table(d);
    d->lines = lines; d->lines_bytes = ht_items * (64ull << (lplog - 2));
    d->ovf = ovf; d->ovf_n = ovf_n;
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
    free_table(d);
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
    free_table(d);
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

// ---- tiles ------------------------------------------------------------------------------------------------
static void le_to_fe(fe &f, const uint8_t *le) { memcpy(f.v, le, 32); }

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
    if (chain_group(d, d->pi) != 4) return nullptr;
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
    const uint32_t group = chain_group(d, pi);
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
    const uint32_t tpl = auto_tiles_per_launch(d);
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
    int rc = ensure_chain(d, auto_tiles_per_launch(d));
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
    le_to_fe(d->walk_p0x, p0_xy_le); le_to_fe(d->walk_p0y, p0_xy_le + 32);
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
    const uint32_t tpl = auto_tiles_per_launch(d);
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
    if (lines_layout(d) && d->lines && d->lines_owned && d->chain_pieces.empty() && d->lines_bytes <= (40ull << 30)) {
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

// ---- probe digest (parity instrumentation): per engine thread, XOR and wrapping sum of every 64-bit key it probed -------
extern "C" int bsgs_run_digest(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles, uint64_t *digest_out, bsgs_hit_ex *hits,
                               uint32_t max_hits, uint32_t *nhits)
{
    if (!d || !centres || !digest_out) return fail(BSGS_ERR_ARG, "null");
    if (!d->g2 || !d->layout) return fail(BSGS_ERR_STATE, "upload giants and table first");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    if ((d->pi & 1u) || !lines_layout(d)) return fail(BSGS_ERR_STATE, "the digest is an instrument of the default (pair-batched, bucket-line) kernel");
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
    if ((d->pi & 1u) || !lines_layout(d)) return fail(BSGS_ERR_STATE, "default kernel only");
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
    auto fail_all = [&](int code) {
        const std::string why = bsgs_last_error();
        for (int k = 1; k < n; k++) { (void)hipSetDevice(devs[k]->id); (void)hipStreamSynchronize(devs[k]->stream); if (what & 2u) { devs[k]->lines_owned = true; devs[k]->csr_owned = true; free_table(devs[k]); } }
        bsgs_fabric_close(F);
        return fail(code, "%s", why.c_str());
    };
    for (int i = 1; i < n; i++) {
        bsgs_dev *d = devs[i];
        auto prepare = [&]() -> int {
            HIPCHK(hipSetDevice(d->id));
            if (what & 1u) {
                int r = set_geometry(d, s->t, s->b, s->p);
                if (r) return r;
                if (d->Ti != s->Ti || d->pi != s->pi) return fail(BSGS_ERR_STATE, "device %d chose another batching", i);
                g2[i] = d->g2;
            }
            if (what & 2u) {
                free_table(d);
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
            d->ht_items = s->ht_items; d->w = s->w; d->overflow = s->overflow; d->lines_bytes = s->lines_bytes; d->layout = s->layout; d->bucket_mul = s->bucket_mul;
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
    int r = set_geometry(d, s->t, s->b, s->p);
    if (r) return r;
    if (d->Ti != s->Ti || d->pi != s->pi) return fail(BSGS_ERR_STATE, "the twin chose another batching");
    HIPCHK(hipStreamSynchronize(s->stream));                          // whatever built the owner's buffers is done
    HIPCHK(hipMemcpyAsync(d->g2, s->g2, s->maxnonce * 64, hipMemcpyDeviceToDevice, d->stream));
    free_table(d);
    d->csr = s->csr; d->csr_owned = false;
    d->lines = s->lines; d->ovf = s->ovf; d->ovf_n = s->ovf_n; d->lines_owned = false;
    d->ht_items = s->ht_items; d->w = s->w; d->overflow = s->overflow; d->lines_bytes = s->lines_bytes; d->layout = s->layout; d->bucket_mul = s->bucket_mul;
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
        if (d->layout == BSGS_TABLE_LINES128) hipLaunchKernelGGL(table_census_kernel<3>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->lines, d->ht_items, (const u32 *)d->csr, (const u64 *)d->ovf, d->ovf_n, c);
        else                                  hipLaunchKernelGGL(table_census_kernel<2>, dim3(blocks), dim3(256), 0, d->stream, (const u32x4 *)d->lines, d->ht_items, (const u32 *)d->csr, (const u64 *)d->ovf, d->ovf_n, c);
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

// test hook: flip bits of ONE byte of the installed table (bucket lines if the layout has them, else the CSR image) -- the corrupted replica
// the verification must catch (tests/test_gpu_round4.py, bench.py BENCH_CORRUPT_RANK)
extern "C" int bsgs_debug_corrupt_table(bsgs_dev *d, uint64_t byte_offset, uint32_t xor_mask)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (!d->layout) return fail(BSGS_ERR_STATE, "no table on device");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued: collect them first");
    uint8_t *base = d->lines ? (uint8_t *)d->lines : (uint8_t *)d->csr;
    const uint64_t bytes = d->lines ? d->lines_bytes : 4 * (d->ht_items + 1) + 4 * d->w;
    if (byte_offset >= bytes) return fail(BSGS_ERR_ARG, "offset %llu beyond the %llu bytes of the table", (unsigned long long)byte_offset, (unsigned long long)bytes);
    HIPCHK(hipSetDevice(d->id));
    uint8_t v = 0;
    HIPCHK(hipMemcpy(&v, base + byte_offset, 1, hipMemcpyDeviceToHost));
    v ^= (uint8_t)xor_mask;
    HIPCHK(hipMemcpy(base + byte_offset, &v, 1, hipMemcpyHostToDevice));
    return BSGS_OK;
}

// ---- phase timing: the same batch run with the kernel stopping after phase 1, after phase 2, and in full ------
extern "C" int bsgs_profile_phases(bsgs_dev *d, const uint8_t *centres, uint32_t ntiles, float ms_out[3])
{
    if (!d || !centres || !ms_out) return fail(BSGS_ERR_ARG, "null");
    if (chain_group(d, d->pi) < 2) return fail(BSGS_ERR_STATE, "phase timing is an instrument of the chained kernel (bucket lines, even batch length)");
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
    le_to_fe(Px, px_le); le_to_fe(Py, py_le);
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

// diagnostics: give ONE of the engine's buffers a new allocation with the same contents (0 = bucket lines, 1 = chain scratch,
// 2 = giants), optionally after a `spacer_bytes` allocation that is released again (so the new one lands elsewhere)
extern "C" int bsgs_debug_realloc(bsgs_dev *d, int which, uint64_t spacer_bytes)
{
    if (!d) return fail(BSGS_ERR_ARG, "null");
    if (d->queued) return fail(BSGS_ERR_STATE, "tiles are queued");
    HIPCHK(hipSetDevice(d->id));
    HIPCHK(hipStreamSynchronize(d->stream));
    void *spacer = nullptr;
    if (spacer_bytes) HIPCHK(hipMalloc(&spacer, spacer_bytes));
    hipError_t e = hipSuccess;
    if (which == 0 && d->lines && d->lines_owned) {
        void *n = nullptr;
        e = bsgs_big_malloc(&n, d->lines_bytes);
        if (e == hipSuccess) e = hipMemcpy(n, d->lines, d->lines_bytes, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) { (void)bsgs_big_free(d->lines); d->lines = (u32x4 *)n; }
    } else if (which == 1 && (d->chain || !d->chain_pieces.empty())) {
        if (d->chain) (void)hipFree(d->chain);
        free_chain_pieces(d);
        d->chain = nullptr; d->chain_bytes = 0;                                             // scratch: the next enqueue allocates it again
    } else if (which == 3) {                                     // a new HIP stream (= possibly another hardware queue)
        hipStream_t ns = nullptr;
        e = hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
        if (e == hipSuccess) { (void)hipStreamDestroy(d->stream); d->stream = ns; }
    } else if (which == 4 && d->hitbuf) {                        // hit buffer + centres
        u32 *nh = nullptr;
        e = hipMalloc(&nh, hitbuf_bytes(d));
        if (e == hipSuccess) e = hipMemset(nh, 0, 64);
        if (e == hipSuccess) { (void)hipFree(d->hitbuf); d->hitbuf = nh; }
        if (d->cen_dev) { (void)hipFree(d->cen_dev); d->cen_dev = nullptr; }
        if (d->cen_pin) { (void)hipHostFree(d->cen_pin); d->cen_pin = nullptr; }
        d->cen_cap = 0;
    } else if (which == 2 && d->g2) {
        void *n = nullptr;
        e = hipMalloc(&n, d->maxnonce * 64);
        if (e == hipSuccess) e = hipMemcpy(n, d->g2, d->maxnonce * 64, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) { (void)hipFree(d->g2); d->g2 = (u32x4 *)n; }
    }
    if (spacer) (void)hipFree(spacer);
    if (e != hipSuccess) return fail(BSGS_ERR_HIP, "debug_realloc: %s", hipGetErrorString(e));
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
