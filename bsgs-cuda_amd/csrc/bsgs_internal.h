// bsgs_internal.h -- private to libbsgs_hip.so: the device object shared by its translation units.
#pragma once
#include "giant_kernel.hip.h"
#include "../../include/bsgs_hip.h"
#include <cstring>
#include <string>
#include <vector>

int bsgs_fail(int code, const char *fmt, ...);
#define HIPCHK(x)                                                                                        \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) return bsgs_fail(BSGS_ERR_HIP, "%s -> %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define fail bsgs_fail

struct bsgs_dev {
    int id = 0;
    hipStream_t stream = nullptr;          // the engine's one stream: uploads, relayouts, tile launches, read-backs
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned debug_flags = 0;
    bool phase_probe = false;
    unsigned block_size = 256;             // threads per workgroup of the tile kernel (four waves: one Fermat inversion per block)
    hipDeviceProp_t prop;
    // geometry
    uint32_t t = 0, b = 0, p = 0;          // the caller's geometry (file layout, hit index i = tid*p + j)
    uint64_t T = 0, maxnonce = 0;
    uint32_t Ti = 0, pi = 0;               // the engine's own: Ti threads x pi giants per inversion, Ti*pi = maxnonce
    uint64_t chain_bytes = 0;              // size of the chain scratch
    // pair-batched kernel, one stream: the scratch in separately allocated, graded pieces of 2^chain_piece_log tiles (ensure_chain)
    std::vector<u32x4 *> chain_pieces;
    uint32_t chain_piece_log = 0;
    uint64_t chain_piece_bytes = 0;
    uint32_t chain_graded = 0, chain_rejected = 0;      // pieces graded / handed back by the last graded allocation
    std::vector<void *> group0_reserve;                 // big tables: pieces of the grader's memory group held back for the chain scratch
    uint64_t group0_piece_bytes = 0;
    uint32_t group0_graded = 0; float group0_grade_lo = 0.f, group0_grade_hi = 0.f;
    unsigned long long *grade_idx = nullptr, *grade_out = nullptr;   // the grader's index / output streams (kept for the engine's life: grades are relative to them)
    hipEvent_t grade_ea = nullptr, grade_eb = nullptr;
    uint32_t chain_from_reserve = 0;
    uint32_t chain_separated = 0;                       // the last graded allocation saw two classes (a piece >= 5 % below the best); 0 = the grades carried no information
    std::vector<float> chain_grades;                    // every grade of the last graded allocation, kept pieces first
    float chain_grade_best = 0.f, chain_grade_worst = 0.f;   // grade (G gathers/s) of the best / worst piece kept
    uint32_t chain_pad = 0;                       // extra u32x4 elements between the scratch areas of consecutive tiles
    std::vector<void *> pending_dev, pending_pinned;   // per-enqueue centre buffers, released by bsgs_collect
    // buffers
    u32x4 *g2 = nullptr;        // [p][4][T]
    // The same giants dealt to MORE threads with shorter batches (giant i = thread * pi + slot for every factorisation Ti * pi = maxnonce, so hit
    // indices do not change): built on demand for launches of so few tiles that the default batching leaves most of the GPU idle -- the reference's own
    // pattern is ONE tile per launch (1_9_7File.pb:2442-2459).  `narrow_off`: BSGS_NARROW_LAUNCHES=0 (A-B), or a layout that did not fit once.
    struct Batching { uint32_t Ti = 0, pi = 0; u32x4 *g2 = nullptr; };
    std::vector<Batching> narrow;
    bool narrow_off = false;               // this geometry's narrow copies did not fit once (reset by set_geometry / a table change)
    bool narrow_env_off = false;           // BSGS_NARROW_LAUNCHES=0
    uint32_t last_Ti = 0, last_pi = 0;     // batching of the last tile launch (bsgs_debug_last_batching)
    u32x4 *chain = nullptr;     // one-buffer scratch: [tile][p][2][T] (per-giant kernel) or the chained kernel's layout when not in pieces
    u32 *csr = nullptr;         // htGPU image
    bool csr_owned = true;
    u32x4 *lines = nullptr;
    bool lines_owned = true;    // false: lines / ovf were handed over by bsgs_install_table_ext_device (borrowed)
    u64 *ovf = nullptr;         // "lines + overflow list" formats (no CSR on the device): hash set of (bucket << 32 | hash)
    uint64_t ovf_n = 0;         // slots (power of two)
    bool bound_copies = true;   // the set holds a COPY of every over-full (and exactly full) line's last word: the direct builder's convention, assumed for tables installed from outside;
                                // false: made from an htGPU image (the line holds real entries only).  Only the census' duplicate count depends on it.
    uint64_t ht_items = 0, w = 0, lines_bytes = 0, overflow = 0;
    uint32_t bucket_mul = 0;    // 0: ht_items is a power of two, bucket = x & (ht_items - 1); else = ht_items: any number of buckets, bucket from 48 bits of the key (giant_kernel.hip.h bucket_mul48; extended tables, 128-byte lines)
    uint32_t layout = 0;        // probe layout: 1 csr, 2 lines64, 3 lines128 (ovf != NULL: reported as 4 / 5)
    u32 *hitbuf = nullptr;      // device
    u32 *hit_host = nullptr;    // pinned mirror
    uint32_t max_hits = 1u << 16;
    uint32_t queued = 0;
    uint32_t tiles_per_launch = 0;         // 0 = automatic (fill the chip: Ti * tiles >= 1024 threads per CU)
    uint32_t auto_tpl = 0;                 // the automatic choice, made once the giants and the table are resident (memory permitting: 4x)
    uint64_t launches = 0;
    int variant = 13;           // BSGS_KERNEL_VARIANT (A-B and tests; all bit-identical): 13 (default) = chained kernel, one stored product per FOUR giants, one probe
                                // in flight per wave (falls back to 10 when the engine's batch length is not a multiple of 4); 10 = one stored product per PAIR, two
                                // probes in flight; 0 = the per-giant kernel (also taken for the CSR layout and for odd batch lengths)
    bool timing_open = false;
    // tile centres of the queued launches, device + pinned staging (grow-only; replaced buffers wait in pending_* for bsgs_collect)
    fe *cen_dev = nullptr;
    uint8_t *cen_pin = nullptr;
    uint64_t cen_cap = 0;                  // tiles
    // device-side tile walk (bsgs_set_walk): P_k = P0 + k*D ; table = affine 2^j * D, j = 0..63
    bool walk_set = false;
    fe walk_p0x, walk_p0y;
    fe *walk_table = nullptr;
    // flags (bsgs_set_flags)
    uint32_t flags = 0;
    u32 *quirk_list = nullptr;             // device: giants whose Gy trips the reference's NEGMODP (BSGS_FLAG_REFERENCE_QUIRKS)
    std::vector<uint32_t> quirk_host;      // the same, sorted, for the hit filter of bsgs_collect
    bool quirk_ready = false;
    u64 *digest = nullptr;                 // bsgs_run_digest: [tile][Ti][2]
    uint64_t digest_bytes = 0;
    // receive buffers of an extended table that arrives by broadcast (bsgs_alloc_table_ext_recv): allocated like the engine's own
    // (bsgs_lines_malloc: tables above 40 GiB get a memory group reserved for the chain scratch), owned by the engine
    void *recv_lines = nullptr, *recv_ovf = nullptr;
    const char *last_kernel = "";          // the tile kernel instantiation of the most recent launch (bsgs_debug_last_kernel)
};

// the big, long-lived device buffers (bucket lines, chain scratch, giants).  BSGS_CONTIGUOUS=1: ask for physically contiguous
// memory first (hipDeviceMallocContiguous), plain hipMalloc when that is refused.
hipError_t bsgs_big_malloc(void **p, size_t bytes);
hipError_t bsgs_big_free(void *p);                           // releases what bsgs_big_malloc / bsgs_lines_malloc / the piece allocators handed out (hipMalloc'ed or chunk-mapped)
hipError_t bsgs_lines_malloc(bsgs_dev *d, void **out, size_t bytes);       // bucket lines: the candidate in the gather-slow memory class (bsgs_hip.hip)
// placement.hip
void free_chain_pieces(bsgs_dev *d);                        // the graded pieces of the chain scratch
void free_reserve(bsgs_dev *d);
void trim_reserve(bsgs_dev *d, size_t keep);                // ... all but `keep` pieces of it                             // the memory group held back for the scratch of a large table
void release_grader(bsgs_dev *d);
void park_release(int device);                              // hand every parked piece of this device back to the driver
uint64_t parked_bytes(int device);                          // bytes this process has parked on the device
hipError_t bsgs_mem_available(size_t *avail, size_t *total);   // hipMemGetInfo of the current device + what is parked there (ours on demand)
bool alloc_graded_pieces(bsgs_dev *d, size_t npieces, uint64_t piece_bytes);     // fills d->chain_pieces; false = not enough memory
template <typename T> static inline hipError_t bsgs_big_malloc(T **p, size_t bytes) { return bsgs_big_malloc((void **)p, bytes); }

// shared between the translation units of the library
// tile_lines64.hip / tile_lines128.hip: the tile-kernel instantiations per bucket-line size (tile_launch.inc)
hipError_t bsgs_launch_tile_lines64(const TileArgs &A, dim3 grid, dim3 block, size_t lds, hipStream_t st, uint32_t group, bool dbg, const char **name);
hipError_t bsgs_launch_tile_lines128(const TileArgs &A, dim3 grid, dim3 block, size_t lds, hipStream_t st, uint32_t group, bool dbg, const char **name);
hipError_t bsgs_launch_tile_lines64_any(const TileArgs &A, dim3 grid, dim3 block, size_t lds, hipStream_t st, uint32_t group, bool dbg, const char **name);
// bsgs_hip.hip (the engine core)
void bsgs_free_table(bsgs_dev *d);
void bsgs_free_recv(bsgs_dev *d);                            // receive buffers that were never installed
int bsgs_set_geometry(bsgs_dev *d, uint32_t t, uint32_t b, uint32_t p);      // (re)allocates the giants for this geometry and picks the engine's own batching
uint32_t bsgs_chain_group(const bsgs_dev *d, uint32_t pi);   // giants per stored running product of the tile kernel a launch with batch length pi takes (4 / 2 chained, 1 per-giant)
uint32_t bsgs_auto_tiles_per_launch(const bsgs_dev *d);      // the launch size in effect
static inline bool bsgs_lines_layout(const bsgs_dev *d) { return d->layout == BSGS_TABLE_LINES64 || d->layout == BSGS_TABLE_LINES128; }
static inline size_t bsgs_hitbuf_bytes(const bsgs_dev *d) { return 64 + (size_t)d->max_hits * 16; }
static inline void bsgs_le_to_fe(fe &f, const uint8_t *le) { memcpy(f.v, le, 32); }
uint64_t bsgs_ovf_slots(uint64_t entries);                   // size of the overflow hash set for `entries` keys (power of two, load <= 1/2)
int bsgs_ovf_fill(bsgs_dev *d, const u64 *list, uint64_t n, u64 *table, uint64_t slots);   // table := hash set of list[0..n)
// hand a finished "lines + overflow list" table to the engine (it becomes the owner of both buffers)
int bsgs_install_lines(bsgs_dev *d, u32x4 *lines, int lplog, u64 *ovf, uint64_t ovf_n, uint64_t ht_items, uint64_t w,
                       uint64_t overflow_buckets);


// startup.hip: the fabric between the engines of one process -- RCCL over xGMI (dlopen'ed on first use) or direct peer copies
struct bsgs_fabric;
int bsgs_fabric_open(bsgs_fabric **f, bsgs_dev *const *devs, int n, uint32_t transport);      // transport: BSGS_TRANSPORT_*
void bsgs_fabric_close(bsgs_fabric *f);
const char *bsgs_fabric_name(const bsgs_fabric *f);
int bsgs_fabric_is_rccl(const bsgs_fabric *f);
int bsgs_fabric_broadcast(bsgs_fabric *f, void *const *bufs, size_t bytes, int root);          // bufs[i] on engine i; every engine's stream drained on return
int bsgs_fabric_allgather(bsgs_fabric *f, void *const *bufs, size_t slice_bytes);              // in place: engine i's slice number i is there already
