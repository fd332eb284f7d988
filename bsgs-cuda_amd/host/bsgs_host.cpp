// bsgs_host.cpp -- C++ host of the MI355X BSGS solver: the reference's `bsgscudaHT_1_9_6file.exe` command line,
// file formats and outputs on top of libbsgs_hip.so's native API (include/bsgs_hip.h).
//
// Mirrors (file:line of /root/reference/1_9_7File.pb): flag parser getprogparam 875-1042 and the checks
// 4412-4472, 4616-4630; start-up constants 4689-4712, 4759-4765; table files Save_HTpacked 3645-3759 /
// Save_Load_Giants 1905-2058 (names and byte layouts kept; built on the GPU when missing); per-GPU driver
// thread cuda() 2095-2553; tile dispenser GetJob 2077-2092; hit resolver checkerThread 3933-4296; checkpoint
// saveCurentCNT 3897-3931 and its restore 4634-4686; per-pubkey loop, progress line and win.txt 4995-5168.
// PureBasic is not available in this image, so the host is C++; INTEGRATION.md shows the PureBasic bindings.
//
// Build: make -C bsgs-cuda_amd host   ->  build/bsgs_mi355x
#include "../../include/bsgs_hip.h"
#include "../csrc/host_secp.h"

#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdarg>
#include <deque>
#include <memory>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <functional>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>

using hs::Affine;
using hs::Scalar;

// ---- SHA1 (configuration fingerprint of currentwork.txt, 1_9_7File.pb:4635-4636) -------------------------------
static std::string sha1_hex(const std::string &msg)
{
    uint32_t h[5] = {0x67452301u, 0xEFCDAB89u, 0x98BADCFEu, 0x10325476u, 0xC3D2E1F0u};
    std::string m = msg;
    const uint64_t bits = (uint64_t)msg.size() * 8;
    m.push_back((char)0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 7; i >= 0; i--) m.push_back((char)(bits >> (8 * i)));
    auto rol = [](uint32_t v, int s) { return (v << s) | (v >> (32 - s)); };
    for (size_t off = 0; off < m.size(); off += 64) {
        uint32_t w[80];
        for (int i = 0; i < 16; i++)
            w[i] = ((uint32_t)(uint8_t)m[off + 4 * i] << 24) | ((uint32_t)(uint8_t)m[off + 4 * i + 1] << 16) |
                   ((uint32_t)(uint8_t)m[off + 4 * i + 2] << 8) | (uint32_t)(uint8_t)m[off + 4 * i + 3];
        for (int i = 16; i < 80; i++) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; i++) {
            uint32_t f, k;
            if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
            else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
            else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
            const uint32_t t = rol(a, 5) + f + e + k + w[i];
            e = d; d = c; c = rol(b, 30); b = a; a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
    }
    char out[41];
    snprintf(out, sizeof out, "%08x%08x%08x%08x%08x", h[0], h[1], h[2], h[3], h[4]);
    return out;
}

// ---- configuration ---------------------------------------------------------------------------------------------------
struct Config {
    uint32_t t = 256, b = 132, p = 400;           // defaults 1_9_7File.pb:181-184
    uint64_t w = 1ull << 25;
    uint32_t htsz = 25;
    std::string devices;                           // -d
    std::string pub = "036d05521c67b9cc1c0ef906b42215c7120c7302c34d9316a2726199bedac50936";   // 1_9_7File.pb:191
    std::string pk = "0x01", pke = "1ffffffffffffffff";                                        // 1_9_7File.pb:197, 210
    bool pke_given = false;
    std::string infile, recovery_file;
    int wt = 180;
    bool onlygen = false;                          // onlygen_1_9_6File0.exe behaviour: build files and exit
    bool cpugen = false;                           // -cpugen: table and giants files built on the host CPU (with -onlygen: no GPU is touched at all)
    uint64_t max_tiles = 0;                        // test hook: stop after this many tiles (0 = unlimited)
    bool ext = false;                              // extended table (bucket lines + overflow list, no HT files); implied by w >= 3069485951
    std::string dir = ".";                         // where table / output files live
    bool verify_replicas = true;                   // several engines: compare table checksums and the hits of one tile across them before searching (-noverify skips)
    bool ref_quirks = false;                       // -refquirks: reproduce the reference kernel's NEGMODP bug bit for bit (BSGS_FLAG_REFERENCE_QUIRKS)
    bool host_centres = false;                     // -hostcentres: tile centres added on the host and uploaded (the reference's way) instead of the device walk
    bool tune = false;                             // -tune: also choose the bucket-line placement by measurement at start-up (bsgs_tune_placement)
    std::string joblog;                            // test hook: log every dispenser / checkpoint event to this file
    uint32_t htsz_arg = 25;                        // what the extended-table entry points take as `htsz`: the exponent, or -- `-htsz` with a fraction, `-buckets` -- the bucket COUNT
    std::string startup = "auto";                  // -startup: how N engines get their replicas -- broadcast | local | allgather | auto (include/bsgs_hip.h BSGS_STARTUP_*)
    std::string transport = "auto";                // -transport: rccl | peer | auto
    int lanes = -1;                                // -lanes: jobs (public keys of -infile) searched side by side, each on its own engine per GPU; -1 = automatic (2 for short jobs)
    bool w_auto = false;                           // -w auto: the table Tune picks for the range given (tune_plan)
    bool file_search = true;                       // -sf (hidden in the reference too, 1_9_7File.pb:907-918; its default is 1, 1_9_7File.pb:178, and so is this host's): htCPU looked up in the file instead of RAM when the file is there at start-up (a table built in this run is still in RAM)
};

static void die(const std::string &msg)
{
    fprintf(stderr, "%s\n", msg.c_str());
    exit(1);
}
static std::string cut_hex(std::string s)
{
    if (s.size() >= 2 && s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) s = s.substr(2);
    for (auto &c : s) c = (char)tolower(c);
    return s;
}

static void usage(const Config &c)
{
    printf(" -t      Number of GPU threads, default %u\n -b      Number of GPU blocks, default %u\n -p      Number of pparam, default %u\n"
           " -d      Select GPU IDs, default all\n-pb      Set single uncompressed/compressed pubkey for searching\n"
           "-pk      Range start from , default %s\n-pke     End range \n-w       Set number of baby items 2^ or decimal representation\n"
           "-htsz    Set number of HashTable 2^ , default %u\n-infile  Set file with pubkey for searching in uncompressed/compressed  format (search sequential)\n"
           "-wl      Set recovery file from which the state will be loaded\n-wt      Set timer for autosaving current state, default every %dseconds\n"
           "-onlygen Generate the table files and exit (onlygen_1_9_6File0.exe)\n-cpugen  Build missing table / giants files on the host CPU; with -onlygen no GPU is touched (the reference`s CPU-only generator)\n-dir     Directory for table files, currentwork.txt and win.txt\n"
           "-ext     Extended baby table built in GPU memory (no HT files); automatic for -w above the reference limit, up to 2^36\n"
           "-noverify    Several GPUs: skip the comparison of the replicas (table checksums, one probe tile) after they were made\n"
           "-refquirks   Reproduce the reference kernel's -Gy borrow bug bit for bit (default: correct arithmetic, finds a superset)\n"
           "-hostcentres Add the tile centres on the host and upload them (default: derived on the GPU from the tile counter)\n"
           "-tune        Time a few placements of the GPU buffers at start-up and keep the fastest (the engine already places them by grade)\n"
           "-startup     Several GPUs: broadcast (GPU 0 holds the table, the others receive it over xGMI), local (every GPU builds / uploads its own),\n"
           "             allgather (extended tables: every GPU builds 1/N of the bucket lines, then all-gather); default: local for extended tables, else broadcast\n"
           "-transport   Several GPUs: rccl | peer (direct peer copies) | auto (RCCL when the GPUs are distinct and librccl loads)\n"
           "-lanes       -infile: public keys searched side by side, each on an engine of its own per GPU (default: 2 when a job is only a launch or two long, else 1)\n"
           "-w auto      The table Tune picks for the range given: the one that minimises table build + worst-case search (a 64-bit range: -w 30 -ext)\n"
           "-buckets     Extended table: the number of buckets itself (any number below 2^32; 64-byte lines up to 12.5 items per bucket, else 128-byte lines), e.g. -w 35 -buckets 3221225472\n",
           c.t, c.b, c.p, c.pk.c_str(), c.htsz, c.wt);
}

static Config parse_args(int argc, char **argv)
{
    Config c;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        for (auto &ch : a) ch = (char)tolower(ch);
        auto next = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "-h") { usage(c); exit(0); }
        else if (a == "-t") { c.t = (uint32_t)atoi(next().c_str()); printf("Number of GPU threads set to #%u\n", c.t); }
        else if (a == "-b") { c.b = (uint32_t)atoi(next().c_str()); printf("Number of GPU blocks set to #%u\n", c.b); }
        else if (a == "-p") { c.p = (uint32_t)atoi(next().c_str()); printf("Number of pparam set to #%u\n", c.p); }
        else if (a == "-d") { c.devices = next(); printf("Used GPU devices #%s\n", c.devices.c_str()); }
        else if (a == "-pb") { c.pub = cut_hex(next()); printf("Pubkey set to %s\n", c.pub.c_str()); }
        else if (a == "-pk") { c.pk = cut_hex(next()); printf("Range begin: 0x%s\n", c.pk.c_str()); }
        else if (a == "-pke") { c.pke = cut_hex(next()); c.pke_given = true; printf("Range end: 0x%s\n", c.pke.c_str()); }
        else if (a == "-w") {                                   // <=32: 2^value (fractional allowed), else decimal  (1009-1022)
            const std::string v = next();
            if (v == "auto" || v == "AUTO") { c.w_auto = true; printf("Items number: chosen for the range (Tune)\n"); continue; }
            const double d = atof(v.c_str());
            // the reference switches to decimal above 32; 33..36 are exponents of the extended table here
            if (d <= 36.0) { c.w = (uint64_t)std::pow(2.0, d); printf("Items number set to 2^%s=%llu\n", v.c_str(), (unsigned long long)c.w); }
            else { c.w = strtoull(v.c_str(), nullptr, 10); printf("Items number set to %llu = 2^%f\n", (unsigned long long)c.w, std::log2((double)c.w)); }
        }
        else if (a == "-htsz") {
            const std::string v = next();
            const double d = atof(v.c_str());
            c.htsz = (uint32_t)d; c.htsz_arg = c.htsz;
            if (d != std::floor(d)) {               // a fraction (as -w takes one, 1_9_7File.pb:1009-1022): extended tables may have any number of buckets
                c.htsz_arg = (uint32_t)std::llround(std::pow(2.0, d));
                printf("HT size set to 2^%s=%u buckets (extended table)\n", v.c_str(), c.htsz_arg);
            } else printf("HT size set to 2^%u\n", c.htsz);
        }
        else if (a == "-buckets") { c.htsz_arg = (uint32_t)strtoull(next().c_str(), nullptr, 10); c.htsz = 0; while ((2ull << c.htsz) <= c.htsz_arg) c.htsz++; printf("HT size set to %u buckets (extended table)\n", c.htsz_arg); }
        else if (a == "-sf") { c.file_search = atoi(next().c_str()) != 0; printf(c.file_search ? "Search in file\n" : "Search in RAM\n"); }
        else if (a == "-lanes") { c.lanes = atoi(next().c_str()); if (c.lanes < 1 || c.lanes > 4) die("-lanes 1..4"); }
        else if (a == "-startup") { c.startup = next(); for (auto &ch : c.startup) ch = (char)tolower(ch); }
        else if (a == "-transport") { c.transport = next(); for (auto &ch : c.transport) ch = (char)tolower(ch); }
        else if (a == "-infile") { c.infile = next(); printf("Will be used file: %s\n", c.infile.c_str()); }
        else if (a == "-wl") { c.recovery_file = next(); printf("Recovery work file: %s\n", c.recovery_file.c_str()); }
        else if (a == "-wt") { c.wt = std::max(30, atoi(next().c_str())); printf("Saving timer every %d seconds\n", c.wt); }
        else if (a == "-onlygen") c.onlygen = true;
        else if (a == "-cpugen") c.cpugen = true;
        else if (a == "-maxtiles") c.max_tiles = strtoull(next().c_str(), nullptr, 10);
        else if (a == "-dir") c.dir = next();
        else if (a == "-ext") c.ext = true;
        else if (a == "-refquirks") c.ref_quirks = true;
        else if (a == "-verifyreplicas") c.verify_replicas = true;
        else if (a == "-noverify") c.verify_replicas = false;
        else if (a == "-hostcentres") c.host_centres = true;
        else if (a == "-tune") c.tune = true;
        else if (a == "-joblog") c.joblog = next();
        else die("Unknown parameter " + a);
    }
    // limits 1_9_7File.pb:4412-4418, 4616-4618
    if (c.w >= 3069485951ull) {                    // beyond the reference's u32 file format: extended device-resident table
        if (c.w > (1ull << 36)) die("-w must be at most 2^36");
        c.ext = true;
        printf("-w above the reference limit 3069485951: extended table in GPU memory, no HT files\n");
    }
    if (c.htsz > 31 || c.htsz < 1) die("-htsz must be 1..31");
    if (c.htsz_arg > 31) {
        if (!(c.htsz_arg & (c.htsz_arg - 1))) { c.htsz = 0; while ((1u << c.htsz) < c.htsz_arg) c.htsz++; c.htsz_arg = c.htsz; }      // a power of two after all
        else { c.ext = true; printf("%u buckets (not a power of two): extended table in GPU memory, no HT files\n", c.htsz_arg); }
    }
    if (c.startup != "auto" && c.startup != "broadcast" && c.startup != "local" && c.startup != "allgather") die("-startup: broadcast | local | allgather | auto");
    if (c.transport != "auto" && c.transport != "rccl" && c.transport != "peer") die("-transport: rccl | peer | auto");
    if (c.p & 1) die("-p must be even");
    if (!c.t || !c.b || !c.p) die("-t -b -p must be non-zero");
    return c;
}

// ---- files ---------------------------------------------------------------------------------------------------------------
// the table images on the host (up to 36 GB): plain allocations that are NOT zero-filled first -- a std::vector's resize writes every byte once before the file read or the
// download from the GPU writes it again, a second and a half for the 12 GiB of a -w 30 run
struct HostBuf {
    uint8_t *p = nullptr;
    uint64_t n = 0;
    HostBuf() {}
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    ~HostBuf() { free(p); }
    // Large images (the 5.4 + 9.6 GB of a -w 30 table) are taken 2 MiB-aligned, offered to transparent huge pages and FIRST-TOUCHED BY SEVERAL THREADS: the kernel
    // clears every page it hands out, and one thread faulting 15 GB in (inside a device-to-host copy or a read()) is most of a 3.4 s "build + bring to the host" stage
    void resize(uint64_t bytes)
    {
        free(p); p = nullptr; n = 0;
        if (!bytes) return;
        const bool big = bytes >= (256ull << 20) && !getenv("BSGS_HOST_NO_PREFAULT");
        p = big ? (uint8_t *)aligned_alloc(2u << 20, (bytes + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1)) : (uint8_t *)malloc(bytes);
        if (!p) { fprintf(stderr, "out of host memory (%llu bytes)\n", (unsigned long long)bytes); exit(1); }
        n = bytes;
        if (!big) return;
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        const unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned q = 0; q < nth; q++) th.emplace_back([this, bytes, q, nth]() {
            const uint64_t lo = bytes / nth * q, hi = q + 1 == nth ? bytes : bytes / nth * (q + 1);
            for (uint64_t o = lo; o < hi; o += 4096) ((volatile uint8_t *)p)[o] = 0;
        });
        for (auto &t : th) t.join();
    }
    void release() { free(p); p = nullptr; n = 0; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    uint64_t size() const { return n; }
    const uint8_t &operator[](uint64_t i) const { return p[i]; }
};
static bool file_has_size(const std::string &path, uint64_t expect)
{
    struct stat st;
    return stat(path.c_str(), &st) == 0 && (uint64_t)st.st_size == expect;
}
static bool read_file(const std::string &path, HostBuf &out, uint64_t expect)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const uint64_t n = (uint64_t)f.tellg();
    if (n != expect) return false;
    out.resize(n);
    f.seekg(0);
    f.read((char *)out.data(), (std::streamsize)n);
    return (bool)f;
}
// written under a temporary name and renamed after a checked flush: a run that ends while the file is being written (a later start-up error, a full disk)
// leaves a `.part` file behind, never a short table under the reference's name
static void write_file(const std::string &path, const void *p, uint64_t n)
{
    const std::string tmp = path + ".part";
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f) die("Can`t create " + tmp);
        f.write((const char *)p, (std::streamsize)n);
        f.flush();
        if (!f) { remove(tmp.c_str()); die("Can`t write " + path + " (" + std::to_string(n) + " bytes): disk full?"); }
    }
    if (rename(tmp.c_str(), path.c_str()) != 0) die("Can`t rename " + tmp);
}

#define CK(call) do { int rc_ = (call); if (rc_ != BSGS_OK) die(std::string("error " #call "-") + std::to_string(rc_) + ": " + bsgs_last_error()); } while (0)

// ---- shared state (the reference's globals *GlobKey / GlobPub / checker() / quit) ------------------------------
struct MiniBsgs {
    unsigned mb = 0;
    std::vector<std::pair<uint64_t, uint32_t>> baby;     // (low 64 bits of x(jG), j), j = 1..2^mb, sorted
    Affine Q;                                            // 2^mb * G
    void build(uint64_t w, unsigned threads);
    size_t lookup(uint64_t x64) const;
    std::vector<uint64_t> find(const Affine &T, uint64_t w) const;     // every b' in [1, w] with x(b'G) = x(T)
};

// what every job of a run reads and nobody writes once the start-up is over: the resolver's tables
struct Tables {
    HostBuf htcpu;
    int htcpu_fd = -1;                            // -sf 1 (the reference's default, isFilesearch 1_9_7File.pb:178): htCPU stays in its FILE, a lookup is two reads (ReadHTpackFile /
                                                  // compareHTpackFile 1_9_7File.pb:3056-3099) -- 9.6 GB of host memory and most of the load time of a -w 30 run saved
    MiniBsgs mini;                                // extended tables: the resolver's own small BSGS instead of htCPU
};
struct Tile { Scalar key; uint64_t index; };          // counter and dispenser index of a tile: centre = walk_p0 + index * PUBADDBIG
struct PendingHit { uint32_t code, idx; Tile tile; };

struct Shared {
    Config cfg;
    uint64_t maxnonce = 0;
    double job_tiles = 0.0;                        // tiles in the range of the current job (0 = unbounded / unknown), and the engines that share it
    int ngpus = 1;
    uint32_t batch_hint = 0;                       // short jobs: tiles per batch (each batch waits for its checker); 0 = a launch per batch
    Scalar center_big, gstep, start, width;      // p*w ; 4*maxnonce*w ; -pk ; pke-pk
    bool end_range = false, past_end = false;
    Affine addpubg, center, pubadd, start_neg;   // -(2w)G ; -(p*w)G ; -(gstep)G ; -(start)G
    Affine realpub, findpub;
    std::mutex job_mutex;
    Scalar glob_key;                              // counter of the next tile to hand out
    uint64_t glob_index = 0;                      // its index: counter = key0 + index * gstep
    Affine walk_p0;                               // centre of tile 0 of this job: Q' - key0*G - C*G (1_9_7File.pb:5056-5064)
    FILE *joblog = nullptr;
    std::mutex chk_mutex;
    std::condition_variable chk_cv;
    std::deque<PendingHit> checker;
    std::atomic<bool> quit{false}, all_done{false};
    std::atomic<uint64_t> steps_done{0}, tiles_done{0};
    std::atomic<uint64_t> hits_pushed{0};                   // hits handed to the checker threads
    std::atomic<uint64_t> hits_checked{0}, checker_ns{0};   // resolver load: false positives cost CPU (a small BSGS each with an extended table)
    std::atomic<int> gpus_finished{0};
    std::mutex done_mutex;
    std::condition_variable done_cv;
    std::mutex inflight_mutex;
    std::vector<Scalar> inflight;                 // per GPU: counter of the oldest tile it has not finished (checkpoint = min, 1_9_7File.pb:3904-3911)
    std::vector<bool> inflight_valid;
    Scalar winkey;
    bool found = false;
    Tables *tab = nullptr;
    int listpos = 1;
    std::string mainpub_hex;
};

// centre of tile `index`: P0 + index * PUBADDBIG (what GetJob accumulates one addition at a time, 1_9_7File.pb:2077-2092)
static Affine tile_centre(const Shared &S, uint64_t index)
{
    if (!index) return S.walk_p0;
    return hs::point_add(S.walk_p0, hs::point_mul(S.pubadd, hs::fe_from_u64(index)));
}

// GetJob for a batch: hand out `n` consecutive tiles (1_9_7File.pb:2077-2092).  Only the COUNTER advances on the host; the
// centres are derived on the GPU from the tile index (bsgs_enqueue_walk), or by tile_centre() under -hostcentres.
static size_t get_jobs(Shared &S, size_t n, std::vector<Tile> &out, int slot = -1)
{
    std::lock_guard<std::mutex> lk(S.job_mutex);
    out.clear();
    Scalar key = S.glob_key;
    uint64_t index = S.glob_index;
    for (size_t i = 0; i < n; i++) {
        // 1_9_7File.pb:2512-2518 tests the counter AFTER the launch: the first tile whose counter exceeds the width is still
        // searched (a tile reaches 2w*maxnonce - p*w below its counter), then the dispenser closes
        if (S.past_end) break;
        if (S.end_range && hs::fe_cmp(key, S.width) > 0) S.past_end = true;
        if (S.cfg.max_tiles && S.tiles_done.load() + out.size() >= S.cfg.max_tiles) break;
        Tile t; t.key = key; t.index = index;
        out.push_back(t);
        key = hs::sc_add(key, S.gstep);
        index++;
    }
    if (out.empty()) return 0;
    S.glob_key = key;
    S.glob_index = index;
    if (slot >= 0) {   // the batch is in flight from the moment it leaves the dispenser (checkpoint = min over GPUs, 1_9_7File.pb:3904-3911)
        std::lock_guard<std::mutex> lk2(S.inflight_mutex);
        S.inflight[slot] = out[0].key; S.inflight_valid[slot] = true;
        if (S.joblog) { fprintf(S.joblog, "take %d %llu %zu %s\n", slot, (unsigned long long)out[0].index, out.size(), hs::fe_to_hex(out[0].key).c_str()); fflush(S.joblog); }
    }
    return out.size();
}

// ---- resolver: checkerThread 1_9_7File.pb:3933-4296 ---------------------------------------------------------------
static int htcpu_lookup_file(int fd, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max)
{
    const uint32_t b = (uint32_t)key64 & (uint32_t)(ht_items - 1), h = (uint32_t)(key64 >> 32);
    uint32_t se[2];
    if (pread(fd, se, 8, (off_t)(4 * (uint64_t)b)) != 8) die("error during loading from file: pos[" + std::to_string(4 * (uint64_t)b) + "] 8b");
    if (se[1] < se[0] || se[1] - se[0] > (1u << 24)) die("htCPU file: bucket " + std::to_string(b) + " is malformed");
    const uint32_t cnt = se[1] - se[0];
    if (!cnt) return 0;
    std::vector<uint32_t> items(2 * (size_t)cnt);
    const off_t at = (off_t)(4 * (ht_items + 1) + 8 * (uint64_t)se[0]);
    if (pread(fd, items.data(), 8 * (size_t)cnt, at) != (ssize_t)(8 * (size_t)cnt)) die("error during loading from file: pos[" + std::to_string((uint64_t)at) + "] " + std::to_string(8 * (uint64_t)cnt) + "b");
    int n = 0;
    for (uint32_t k = 0; k < cnt; k++) if (items[2 * k] == h) { if (n < max) pos[n] = items[2 * k + 1]; n++; }
    return n;
}
static int htcpu_lookup(const HostBuf &img, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max)
{
    const uint32_t b = (uint32_t)key64 & (uint32_t)(ht_items - 1), h = (uint32_t)(key64 >> 32);
    uint32_t lo, hi;
    memcpy(&lo, &img[4 * (uint64_t)b], 4); memcpy(&hi, &img[4 * ((uint64_t)b + 1)], 4);
    const uint8_t *items = img.data() + 4 * (ht_items + 1);
    int n = 0;
    for (uint32_t k = lo; k < hi; k++) {
        uint32_t v; memcpy(&v, items + 8 * (uint64_t)k, 4);
        if (v == h) { if (n < max) memcpy(&pos[n], items + 8 * (uint64_t)k + 4, 4); n++; }
    }
    return n;
}

// Extended tables have no htCPU (positions): the baby index b' of a hit, x(b'G) = x(T), 1 <= b' <= w, is found by a small
// BSGS of its own: 2^mb stored multiples of G, then T -+ i*(2^mb G) for i <= w / 2^mb, normalised in batches.
size_t MiniBsgs::lookup(uint64_t x64) const
{
    auto it = std::lower_bound(baby.begin(), baby.end(), std::make_pair(x64, (uint32_t)0));
    return (it != baby.end() && it->first == x64) ? (size_t)(it - baby.begin()) : (size_t)-1;
}
void MiniBsgs::build(uint64_t w, unsigned threads)
{
    unsigned lw = 0; while ((1ull << lw) < w) lw++;
    mb = std::min(24u, std::max(8u, lw / 2 + 7));          // 2^24 stored multiples at -w 34: 2 x 1024 batched additions (1 ms) per reported hit; 2^22 (round 2): 4 ms
    const uint64_t M = 1ull << mb;
    baby.resize(M);
    Q = hs::point_mul(hs::G, hs::fe_from_u64(M));
    threads = std::max(1u, std::min(threads, 64u));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++) th.emplace_back([&, t]() {
        const uint64_t lo = 1 + M * t / threads, hi = 1 + M * (t + 1) / threads;          // j in [lo, hi)
        hs::Jac cur = hs::to_jac(hs::point_mul(hs::G, hs::fe_from_u64(lo)));
        std::vector<hs::Jac> blk;
        for (uint64_t j = lo; j < hi;) {
            blk.clear();
            const uint64_t n = std::min<uint64_t>(4096, hi - j);
            for (uint64_t k = 0; k < n; k++) { blk.push_back(cur); cur = hs::jac_add_affine(cur, hs::G); }
            const std::vector<Affine> a = hs::batch_to_affine(blk);
            for (uint64_t k = 0; k < n; k++) baby[j - 1 + k] = {a[k].x.l[0], (uint32_t)(j + k)};
            j += n;
        }
    });
    for (auto &x : th) x.join();
    std::sort(baby.begin(), baby.end());
}
std::vector<uint64_t> MiniBsgs::find(const Affine &T, uint64_t w) const
{
    std::vector<uint64_t> cand, out;
    const uint64_t M = 1ull << mb, I = w / M + 1;
    const Affine nQ = hs::affine_neg(Q);
    hs::Jac up = hs::to_jac(T), dn = hs::to_jac(T);
    std::vector<hs::Jac> blk;
    std::vector<uint64_t> idx;
    for (uint64_t i = 0; i <= I;) {
        blk.clear(); idx.clear();
        for (int k = 0; k < 256 && i <= I; k++, i++) {
            blk.push_back(up); idx.push_back(i);
            if (i) { blk.push_back(dn); idx.push_back(i); }
            up = hs::jac_add_affine(up, Q); dn = hs::jac_add_affine(dn, nQ);
        }
        const std::vector<Affine> a = hs::batch_to_affine(blk);
        for (size_t k = 0; k < a.size(); k++) {
            const uint64_t base = idx[k] * M;
            if (a[k].inf) { cand.push_back(base); continue; }
            const size_t at = lookup(a[k].x.l[0]);
            if (at == (size_t)-1) continue;
            for (size_t q = at; q < baby.size() && baby[q].first == a[k].x.l[0]; q++) {
                cand.push_back(base + baby[q].second);
                if (base >= baby[q].second) cand.push_back(base - baby[q].second);
            }
        }
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    for (uint64_t b : cand) {
        if (b < 1 || b > w) continue;
        const Affine v = hs::point_mul(hs::G, hs::fe_from_u64(b));
        if (!v.inf && hs::fe_equal(v.x, T.x)) out.push_back(b);
    }
    return out;
}

static bool try_key(const Shared &S, const Scalar &kprime, Scalar &key_out)
{
    const Affine tp = hs::point_mul(hs::G, kprime);
    if (tp.inf || !hs::fe_equal(tp.x, S.findpub.x) || !hs::fe_equal(tp.y, S.findpub.y)) return false;
    const Scalar key = hs::sc_add(kprime, S.start);
    const Affine rp = hs::point_mul(hs::G, key);
    if (rp.inf || !hs::fe_equal(rp.x, S.realpub.x) || !hs::fe_equal(rp.y, S.realpub.y)) return false;
    key_out = key;
    return true;
}

static bool resolve_hit(const Shared &S, const PendingHit &hit, Scalar &key_out)
{
    // k' = cnt + C + e1*(idx+1)*2w + e2*b'   (all sign pairs are verified by scalar multiplication)
    const Scalar base = hs::sc_add(hit.tile.key, S.center_big);
    const Scalar two_w = hs::sc_from_u128((hs::u128)S.cfg.w * 2);
    const Scalar g = hit.code == 5 ? hs::fe_from_u64(0) : hs::sc_mul_small(two_w, (uint64_t)hit.idx + 1);
    if (hit.code == 4) {
        Scalar k = hs::sc_add(base, g); if (try_key(S, k, key_out)) return true;
        k = hs::sc_sub(base, g); return try_key(S, k, key_out);
    }
    const Affine centre = tile_centre(S, hit.tile.index);
    Affine T = centre;
    if (hit.code != 5) {
        Affine gi = hs::point_mul(S.addpubg, hs::fe_from_u64((uint64_t)hit.idx + 1));
        if (hit.code == 2) gi = hs::affine_neg(gi);
        T = hs::point_add(centre, gi);
        if (T.inf) return false;
    }
    std::vector<uint64_t> babies;                 // b' with x(b'G) = x(T) as far as the table knows
    if (S.cfg.ext) babies = S.tab->mini.find(T, S.cfg.w);
    else {
        uint32_t pos[64];
        int np = S.tab->htcpu_fd >= 0 ? htcpu_lookup_file(S.tab->htcpu_fd, 1ull << S.cfg.htsz, T.x.l[0], pos, 64) : htcpu_lookup(S.tab->htcpu, 1ull << S.cfg.htsz, T.x.l[0], pos, 64);
        for (int q = 0; q < std::min(np, 64); q++) babies.push_back((uint64_t)pos[q] + 1);
    }
    for (uint64_t bprime : babies) {
        const Scalar bb = hs::fe_from_u64(bprime);
        for (int s1 = 0; s1 < 2; s1++) {
            Scalar e1g;
            if (hit.code == 5) { if (s1) break; e1g = base; }
            else e1g = ((hit.code == 1) ^ (s1 == 1)) ? hs::sc_add(base, g) : hs::sc_sub(base, g);
            Scalar k = hs::sc_add(e1g, bb); if (try_key(S, k, key_out)) return true;
            k = hs::sc_sub(e1g, bb); if (try_key(S, k, key_out)) return true;
        }
    }
    return false;
}

static void checker_thread(Shared *S)
{
    for (;;) {
        PendingHit hit;
        {
            std::unique_lock<std::mutex> lk(S->chk_mutex);
            S->chk_cv.wait(lk, [&] { return !S->checker.empty() || S->all_done.load(); });
            if (S->checker.empty()) return;
            hit = S->checker.front();
            S->checker.pop_front();
        }
        if (S->quit.load()) continue;
        Scalar key;
        const auto tc0 = std::chrono::steady_clock::now();
        const bool solved = resolve_hit(*S, hit, key);
        S->checker_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tc0).count();
        if (solved) {
            std::lock_guard<std::mutex> lk(S->chk_mutex);
            S->winkey = key; S->found = true;
            S->quit.store(true);
        }
        S->hits_checked++;                      // after `quit`: a driver thread that waits for its hits to be resolved (short jobs) sees the verdict with the count
    }
}

// ---- per-GPU driver thread: cuda() 1_9_7File.pb:2095-2553 ---------------------------------------------------------
// devices are opened and loaded once (1_9_7File.pb:2181-2357) and serve every public key of the run.  Only the first device
// takes the giants and the table from the host (or builds the extended table); the others receive replicas device-to-device
// (bsgs_broadcast_tables) instead of the reference's per-GPU upload over PCIe (1_9_7File.pb:2337, 2350).
static bsgs_dev *open_dev(int gpu)
{
    bsgs_dev *dev = nullptr;
    CK(bsgs_dev_open(gpu, &dev));
    char name[256];
    CK(bsgs_dev_name(dev, name, sizeof name));
    uint64_t fr = 0, tot = 0;
    CK(bsgs_dev_meminfo(dev, &fr, &tot));
    printf("GPU #%d %s memory %.0f/%.0f MB\n", gpu, name, fr / 1048576.0, tot / 1048576.0);
    return dev;
}
// the extended table's line size: 64-byte lines up to 12.5 entries per bucket on average -- at load 8 (-w 34 -htsz 31) one line in 120 is over-full, at 10.67 one in 13,
// and the probes that go on to the overflow set cost 4.3 % (load 10.67) to 7 % (load 12) on 128 GiB of lines (39.6 -> 37.9 -> 36.8 G, profiles/r07m_fuller_lines.log), still
// level with or ahead of the 128-byte-line kernel on the same bytes of table (36.7 G at 1.5 * 2^34 items, 35.7 G at 2^35 where the 64-byte lines do 36.9 G: r07m, r07n) --,
// 128-byte lines beyond that (up to ~24 per bucket) when they fit
static uint32_t ext_layout(const Config &c, uint64_t free_bytes)
{
    const uint64_t buckets = c.htsz_arg > 31 ? c.htsz_arg : 1ull << c.htsz_arg;
    const double load = (double)c.w / (double)buckets;
    const bool fits128 = 128ull * buckets + (24ull << 30) < free_bytes;
    return load > 12.5 && fits128 ? BSGS_TABLE_LINES128_LIST : BSGS_TABLE_LINES64_LIST;
}
static uint32_t transport_code(const Config &c) { return c.transport == "rccl" ? BSGS_TRANSPORT_RCCL : c.transport == "peer" ? BSGS_TRANSPORT_PEER : BSGS_TRANSPORT_AUTO; }
static const char *transport_name(uint32_t t) { return t == BSGS_TRANSPORT_RCCL ? "RCCL over xGMI" : t == BSGS_TRANSPORT_PEER ? "peer copies" : "none"; }
static void print_placement(int gpu, size_t gi, bsgs_dev *dev)
{
    uint32_t info[5] = {0, 0, 0, 0, 0}; float grade[2] = {0.f, 0.f};
    CK(bsgs_chain_placement(dev, info, grade));
    printf("GPU #%d engine %zu: chain scratch in %u piece(s) of %u tiles, %u graded, reserved group: %s\n", gpu, gi, info[0], info[1], info[2], info[4] ? "yes" : "no");
}
// fn(gi) for every engine, one host thread per GPU: engines on distinct GPUs run concurrently, engines that share a GPU (-d 0,0) one after the other
static void per_gpu(const std::vector<int> &gpus, const std::function<void(size_t)> &fn)
{
    std::vector<std::thread> th;
    for (size_t i = 0; i < gpus.size(); i++) {
        bool first = true;
        for (size_t j = 0; j < i; j++) first &= gpus[j] != gpus[i];
        if (!first) continue;
        th.emplace_back([&, i] { for (size_t k = i; k < gpus.size(); k++) if (gpus[k] == gpus[i]) fn(k); });
    }
    for (auto &t : th) t.join();
}

// Devices are loaded once (1_9_7File.pb:2181-2357) and serve every public key of the run.  The reference gives every GPU its own upload of the two host buffers over
// PCIe (1_9_7File.pb:2337, 2350, 4769-4843).  Here, with several engines (-startup):
//   broadcast  engine 0 takes the giants and the table from the host (or builds the extended table), the others receive replicas over xGMI (RCCL, or peer copies);
//   local      every engine takes / builds its own, concurrently: the reference's shape for file tables, and NO link traffic at all for extended tables (default there);
//   allgather  extended tables: every engine builds the lines of 1/N of the buckets, then all-gather.
// Every engine allocates its chain scratch (placed by grade: the reference's cuMemAlloc_v2 before its loop, 1_9_7File.pb:2251) right after its table.
static void load_engines(const Shared &S, const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs, const HostBuf &htgpu, const HostBuf &g2)
{
    const Config &c = S.cfg;
    const size_t n = devs.size();
    const auto t0 = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    std::string strategy = c.startup;
    if (strategy == "auto") strategy = c.ext ? "local" : "broadcast";
    if (!c.ext && strategy == "allgather") { printf("-startup allgather applies to extended tables: file tables are broadcast\n"); strategy = "broadcast"; }
    if (n == 1) strategy = "local";
    const bool local = strategy == "local";
    // ---- giants
    if (local) per_gpu(gpus, [&](size_t gi) { CK(bsgs_upload_g2(devs[gi], g2.data(), c.t, c.b, c.p)); });
    else {
        CK(bsgs_upload_g2(devs[0], g2.data(), c.t, c.b, c.p));
        uint32_t used = 0; double s = 0.0;
        CK(bsgs_broadcast_tables_ex(devs.data(), (int)n, transport_code(c), 1u, &used, &s));
        printf("Giants replicated to %zu more GPU engine(s) by %s in %.2fs\n", n - 1, transport_name(used), s);
    }
    printf("[startup] %-44s %.3fs\n", "giants on every engine", secs());
    // ---- table
    if (c.ext) {
        uint64_t fr = 0, tot = 0;
        CK(bsgs_dev_meminfo(devs[0], &fr, &tot));
        const uint32_t layout = ext_layout(c, fr);
        const uint32_t strat = strategy == "broadcast" ? BSGS_STARTUP_BROADCAST : strategy == "allgather" ? BSGS_STARTUP_ALLGATHER : BSGS_STARTUP_LOCAL;
        std::vector<bsgs_startup_report> rep(n);
        CK(bsgs_startup_ext_tables(devs.data(), (int)n, c.w, c.htsz_arg, layout, strat, transport_code(c), rep.data()));
        static const char *names[3] = {"broadcast", "local", "allgather"};
        for (size_t gi = 0; gi < n; gi++) {
            const bsgs_startup_report &r = rep[gi];
            printf("[startup] engine %zu (GPU #%d) extended table, strategy %s%s: buffers %.2fs, build %.2fs, transfer %.2fs (%.1f GiB received, %s), overflow set %.2fs, install %.2fs, "
                   "chain scratch %.2fs; done at %.2fs\n", gi, gpus[gi], names[r.strategy], r.strategy != strat ? " (fallback)" : "", r.alloc_s, r.build_s, r.transfer_s,
                   r.bytes_received / 1073741824.0, transport_name(r.transport), r.set_s, r.install_s, r.prepare_s, r.total_s);
        }
        uint32_t lay = 0; uint64_t bytes = 0, ovf = 0;
        CK(bsgs_table_info(devs[0], &lay, &bytes, &ovf));
        printf("Extended table: %llu items in %llu lines of %d bytes, %.1f GiB in memory per GPU, %llu over-full buckets, %zu engine(s) ready in %.1fs\n", (unsigned long long)c.w,
               (unsigned long long)(c.htsz_arg > 31 ? c.htsz_arg : 1ull << c.htsz_arg), layout == BSGS_TABLE_LINES128_LIST ? 128 : 64, bytes / 1073741824.0, (unsigned long long)ovf, n, secs());
        for (size_t gi = 0; gi < n; gi++) print_placement(gpus[gi], gi, devs[gi]);
    } else if (local) {
        per_gpu(gpus, [&](size_t gi) {
            CK(bsgs_upload_htgpu(devs[gi], htgpu.data(), 1ull << c.htsz, c.w, BSGS_TABLE_AUTO));
            CK(bsgs_prepare(devs[gi]));
        });
        if (n > 1) printf("Tables uploaded to every GPU engine from the host (the reference's way, 1_9_7File.pb:2337, 2350) in %.2fs\n", secs());
        for (size_t gi = 0; gi < n; gi++) print_placement(gpus[gi], gi, devs[gi]);
    } else {
        CK(bsgs_upload_htgpu(devs[0], htgpu.data(), 1ull << c.htsz, c.w, BSGS_TABLE_AUTO));
        // the first engine's chain scratch BEFORE the replicas: an engine that reserved a memory group for it (tables above 40 GiB) hands the unused part back
        // here, which matters when a second engine shares the GPU (-d 0,0)
        CK(bsgs_prepare(devs[0]));
        print_placement(gpus[0], 0, devs[0]);
        uint32_t used = 0; double s = 0.0;
        CK(bsgs_broadcast_tables_ex(devs.data(), (int)n, transport_code(c), 2u, &used, &s));
        uint32_t lay = 0; uint64_t bytes = 0, ovf = 0;
        CK(bsgs_table_info(devs[0], &lay, &bytes, &ovf));
        printf("Tables replicated to %zu more GPU engine(s) by %s in %.2fs (%.2f GiB each, %.1f GB/s per destination)\n", n - 1, transport_name(used), s, bytes / 1073741824.0,
               s > 0 ? bytes / 1e9 / s : 0.0);
        for (size_t gi = 1; gi < n; gi++) { CK(bsgs_prepare(devs[gi])); print_placement(gpus[gi], gi, devs[gi]); }
    }
    printf("[startup] %-44s %.3fs\n", (std::string("tables on every engine (") + strategy + ")").c_str(), secs());
}

// A replica that differs from the first engine's tables in one byte loses keys silently.  The reference uploads every GPU from ONE host buffer
// (1_9_7File.pb:2337, 2350, 4769-4843); ours travelled device-to-device, so they are compared before the search starts: the 64-bit checksums
// each engine computes over what it holds (bsgs_table_checksum), and the complete hit list of one probe tile run on every engine.
static void verify_replicas(const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs)
{
    if (const char *e = getenv("BSGS_TEST_CORRUPT_ENGINE")) {         // test hook: one flipped bit in one engine's table must stop the run
        const size_t k = (size_t)atoi(e);
        if (k < devs.size()) {
            fprintf(stderr, "BSGS_TEST_CORRUPT_ENGINE=%zu: TEST HOOK -- one bit of engine %zu's table is flipped before the replicas are compared (this run must stop)\n", k, k);
            CK(bsgs_debug_corrupt_table(devs[k], 4096 + 5, 0x10));
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::array<uint64_t, 4>> sums(devs.size());
    for (size_t gi = 0; gi < devs.size(); gi++) CK(bsgs_table_checksum(devs[gi], sums[gi].data()));
    uint8_t centre[64];
    hs::affine_to_le(hs::point_mul(hs::G, hs::fe_from_u64(0x5EEDC0FFEEull)), centre, centre + 32);
    std::vector<std::vector<bsgs_hit_ex>> hits(devs.size(), std::vector<bsgs_hit_ex>(65536));
    std::vector<uint32_t> nh(devs.size(), 0);
    for (size_t gi = 0; gi < devs.size(); gi++) {
        const int rc = bsgs_run(devs[gi], centre, 1, hits[gi].data(), (uint32_t)hits[gi].size(), &nh[gi], nullptr);
        if (rc != BSGS_OK && rc != BSGS_ERR_OVERFLOW) die(std::string("replica verification: ") + bsgs_last_error());
        hits[gi].resize(std::min<uint32_t>(nh[gi], 65536));
    }
    for (size_t gi = 1; gi < devs.size(); gi++) {
        if (sums[gi] != sums[0]) {
            static const char *what[4] = {"bucket lines", "overflow set", "htGPU image", "giants"};
            for (int k = 0; k < 4; k++) if (sums[gi][k] != sums[0][k])
                fprintf(stderr, "GPU #%d engine %zu: checksum of the %s is %016llx, engine 0 has %016llx\n", gpus[gi], gi, what[k], (unsigned long long)sums[gi][k], (unsigned long long)sums[0][k]);
            die("replica verification FAILED: the tables of GPU #" + std::to_string(gpus[gi]) + " differ from the first engine's");
        }
        if (nh[gi] != nh[0] || memcmp(hits[gi].data(), hits[0].data(), hits[0].size() * sizeof(bsgs_hit_ex)) != 0)
            die("replica verification FAILED: GPU #" + std::to_string(gpus[gi]) + " reports other hits than the first engine for the same tile");
    }
    printf("Replica verification: %zu engines hold identical tables (lines %016llx, overflow set %016llx, image %016llx, giants %016llx), probe tile: %u hits on each, in %.2fs\n",
           devs.size(), (unsigned long long)sums[0][0], (unsigned long long)sums[0][1], (unsigned long long)sums[0][2], (unsigned long long)sums[0][3], nh[0],
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
}

static void gpu_thread(Shared *S, int gpu, int slot, bsgs_dev *dev)
{
    uint32_t tpl = 48;
    if (bsgs_tiles_per_launch(dev, &tpl) != BSGS_OK || !tpl) tpl = 48;
    // One batch = one launch (the engine's choice: 48..192 tiles); a found key stops the job at the next batch boundary.  A job that is only a launch or two
    // long (BASELINE config 4: a 64-bit range at -w 30 is 129 tiles) would always run to its end that way -- the reference, one tile per launch, stops at the
    // hit (1_9_7File.pb:2442-2523) -- so such a job is dealt in about six batches per GPU (not below 16 tiles: the narrow batchings keep small launches at
    // 36-38 G): with the key anywhere in the range 0.6 of the work is done on average instead of all of it.
    size_t batch = tpl;
    const bool wait_for_checker = S->batch_hint != 0;
    if (S->batch_hint) batch = std::min<size_t>(S->batch_hint, tpl);
    std::vector<Tile> tiles;
    std::vector<uint8_t> centres;
    std::vector<bsgs_hit_ex> hits(65536);
    auto push_hits = [&](const bsgs_hit_ex *h, uint32_t n, const Tile *base) {
        if (!n) return;
        std::lock_guard<std::mutex> lk(S->chk_mutex);
        for (uint32_t i = 0; i < n; i++) S->checker.push_back({h[i].code, h[i].idx, base[h[i].tile]});
        S->hits_pushed += n;
        S->chk_cv.notify_all();
    };
    // tiles [i0, i0 + n) of the current batch with centres added on the host and uploaded (the reference's way: -hostcentres, and the
    // fallback when the device walk meets the point at infinity)
    auto run_host_centres = [&](size_t i0, size_t n, uint32_t *nh) {
        centres.resize(n * 64);
        for (size_t i = 0; i < n; i++) {
            const Affine c = tile_centre(*S, tiles[i0 + i].index);
            if (c.inf) die("tile centre is the point at infinity (the public key equals -(counter + p*w)*G): the reference cannot search this tile either");
            hs::affine_to_le(c, &centres[i * 64], &centres[i * 64 + 32]);
        }
        return bsgs_run(dev, centres.data(), (uint32_t)n, hits.data(), (uint32_t)hits.size(), nh, nullptr);
    };
    while (!S->quit.load()) {
        const size_t n = get_jobs(*S, batch, tiles, slot);
        if (!n) break;                                            // end of space for this GPU
        uint32_t nh = 0;
        int rc = S->cfg.host_centres ? run_host_centres(0, n, &nh)
                                     : bsgs_run_walk(dev, tiles[0].index, (uint32_t)n, hits.data(), (uint32_t)hits.size(), &nh, nullptr);
        if (rc == BSGS_ERR_DEGENERATE) rc = run_host_centres(0, n, &nh);
        if (rc == BSGS_ERR_OVERFLOW) {
            // more hits than the buffers hold (a degenerate table: tiny -htsz with a large -w): nothing may be dropped silently --
            // the true hit could be among the lost records.  Re-run the batch tile by tile.
            fprintf(stderr, "\nGPU#%d: %u hits in one batch of %zu tiles exceed the hit buffer; re-running tile by tile\n", gpu, nh, n);
            for (size_t i = 0; i < n; i++) {
                uint32_t n1 = 0;
                int r1 = S->cfg.host_centres ? run_host_centres(i, 1, &n1) : bsgs_run_walk(dev, tiles[i].index, 1, hits.data(), (uint32_t)hits.size(), &n1, nullptr);
                if (r1 == BSGS_ERR_DEGENERATE) r1 = run_host_centres(i, 1, &n1);
                if (r1 != BSGS_OK) die(std::string("error bsgs_run-") + std::to_string(r1) + ": " + bsgs_last_error() + " (one tile alone overflows the hit buffer: raise -htsz)");
                push_hits(hits.data(), n1, &tiles[i]);
            }
        } else if (rc != BSGS_OK) die(std::string("error bsgs_run-") + std::to_string(rc) + ": " + bsgs_last_error());
        else push_hits(hits.data(), nh, tiles.data());
        S->steps_done += 2 * S->maxnonce * n;
        S->tiles_done += n;
        // a short job (batches smaller than a launch: see above) does not run ahead of its checker: the next batch is dispensed once this one's hits are resolved
        // (microseconds each with the htCPU table), so that the batch that holds the key is the last one
        if (wait_for_checker) while (!S->quit.load() && S->hits_checked.load() < S->hits_pushed.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
        {
            std::lock_guard<std::mutex> lk(S->inflight_mutex);
            S->inflight_valid[slot] = false;
            if (S->joblog) { fprintf(S->joblog, "done %d %llu %zu\n", slot, (unsigned long long)tiles[0].index, n); fflush(S->joblog); }
        }
    }
    printf("GPU#%d job finished\n", gpu);
    { std::lock_guard<std::mutex> lk(S->done_mutex); S->gpus_finished++; }
    S->done_cv.notify_all();
}

// ---- Tune (1_9_7File.pb:324-431 prints suggested -t -b -p -w -htsz per GPU from free memory and SM count) ----------
// MI355X version: the engine re-batches internally, so -t/-b/-p only set the tile size; -w / -htsz follow from HBM:
// device bytes = 64*2^htsz (bucket lines) + 4*2^htsz + 4*w (htGPU image) + 64*t*b*p (giants) + chain scratch (~8 GiB).
struct TuneAdvice { double w_log2; uint32_t htsz; bool ext; uint32_t ext_w_log2, ext_htsz; };
static TuneAdvice tune_advice(uint64_t free_bytes)
{
    TuneAdvice a{};
    const uint64_t budget = free_bytes > (24ull << 30) ? free_bytes - (24ull << 30) : free_bytes / 2;      // giants, chain scratch, hit buffers, slack
    uint32_t htsz = 20;
    while (htsz < 31 && (68ull << (htsz + 1)) + (16ull << (htsz + 1)) <= budget) htsz++;     // lines + image at 4 entries per bucket
    double wl = htsz + 2.0;                                                          // mean bucket load 4
    const double wmax = std::log2(3069485950.0);                                     // reference format limit (1_9_7File.pb:4412-4418)
    if (wl > wmax) wl = wmax;
    a.w_log2 = wl; a.htsz = htsz;
    // beyond the reference's table format (no HT files): 64-byte bucket lines at 8 entries per bucket, built in GPU memory
    uint32_t eh = 20;
    while (eh < 31 && (64ull << (eh + 1)) <= budget) eh++;
    a.ext = eh + 3 > 31; a.ext_w_log2 = std::min(eh + 3, 36u); a.ext_htsz = eh;
    return a;
}
// Tune for a RANGE.  What a search of 2^range_bits keys costs with w baby steps on n GPUs: the table has to be built (and, in the reference's
// format, brought to the host: the resolver's htCPU and the two HT files), then at most 2^range_bits / (2w * rate * n) seconds are searched -- a small range wants a small
// table, a large one the largest that fits.  Rates measured on MI355X: reference-format build 8.2 G points/s, extended build 11 G/s (10 G/s into 128-byte
// lines), tile kernel 40 G giant-steps/s on 64-byte lines (36 G when the whole job is a launch of < 48 tiles), 33 G on 128-byte lines; 25 GB/s to the host.
struct TunePlan { double w_log2; uint32_t htsz_arg; bool ext; double build_s, search_s, total_s; uint64_t w; };      // w = the number of baby points itself (it need not be a power of two)
static TunePlan tune_plan(uint64_t free_bytes, double range_bits, int n_gpus, uint64_t maxnonce)
{
    const double budget = (double)free_bytes - std::min(34.0 * 1073741824.0, 0.5 * (double)free_bytes);     // chain scratch (24 GiB at most; the engine sizes its launches by what is left), giants, the builder's own scratch
    const double range = std::pow(2.0, range_bits), n = std::max(1, n_gpus);
    TunePlan best{};
    best.total_s = 1e300;
    auto consider = [&](double wl, uint32_t htsz_arg, bool ext, double bytes, double build_rate, double step_rate, double to_host_bytes) {
        if (bytes > budget) return;
        const double w = wl > 36.5 ? wl : std::pow(2.0, wl);                                     // (above 36: the count itself, as -w takes it)
        if (wl > 36.5) wl = std::log2(w);
        const double tiles = std::ceil(range / (4.0 * (double)maxnonce * w)) + 1.0;              // the tile that holds the end of the range is still searched (1_9_7File.pb:2512-2518)
        const double rate = tiles / n < 48.0 ? std::min(step_rate, 36e9) : step_rate;
        TunePlan p{wl, htsz_arg, ext, w / build_rate + to_host_bytes / 25e9, tiles * 2.0 * (double)maxnonce / rate / n, 0.0, (uint64_t)std::llround(w)};
        p.total_s = p.build_s + p.search_s;
        if (p.total_s < best.total_s * 0.999) best = p;
    };
    for (int k = 20; k <= 31; k++) {                                             // the reference's format: 2^(k-2) buckets (load 4), lines + image on the device, both files on the host
        const double w = std::pow(2.0, k), b = std::pow(2.0, k - 2);
        consider(k, (uint32_t)(k - 2), false, 68.0 * b + 4.0 * w, 8.2e9, 40e9, 12.0 * w);
    }
    for (int k = 24; k <= 34; k++) consider(k, (uint32_t)(k - 3), true, 64.0 * std::pow(2.0, k - 3) + 0.04 * std::pow(2.0, k), 11e9, k >= 33 ? 39e9 : 40e9, 0.0);      // extended, 64-byte lines, load 8
    consider(35.0, 1610612736u, true, 128.0 * 1610612736.0 + 5.0 * 1073741824.0, 8.5e9, 35.7e9, 0.0);                                                               // 1.5 * 2^30 lines of 128 bytes (load 21.3 of 30)
    consider(35.0, 3221225472u, true, 64.0 * 3221225472.0 + 17.0 * 1073741824.0, 5.5e9, 38.5e9, 0.0);                                                                // 3 * 2^30 lines of 64 bytes (load 10.67 of 14) + a 16 GiB overflow set: 38.5-38.9 G with the overflow fingerprint in the line headers (r08c, r08d); before it 36.7-37.1 G against 35.7-35.8 G on one box (profiles/r07n_*)
    // 36 * 2^30 points on the same 3 * 2^30 lines (load 12 of 14; 15.6 % of the lines over-full, a 32 GiB overflow set -- the largest count whose set still has 2^32 slots):
    // 37.85 G giant-steps/s against 38.5 G at 2^35, each step covering 12.5 % more keys: 2.93e21 keys/s against 2.65e21 (profiles/r08g_more_points_same_lines.log)
    // (build rates of the two large tables: the WALL the host spends -- 8.0 s for 36 * 2^30 points, of which 3.0 s are the builder's kernels, 2.5-3.9 s one hipMalloc of 192 GiB
    // on a driver that clears what it hands out, the rest the overflow set and the validation: profiles/r08t_config3_key_near_the_start.json, r08t_builder_stages.log)
    consider(38654705664.0, 3221225472u, true, 64.0 * 3221225472.0 + 33.0 * 1073741824.0, 4.8e9, 37.8e9, 0.0);
    if (best.total_s > 1e299) { best = TunePlan{20.0, 18u, false, 0.0, 0.0, 0.0, 1ull << 20}; }
    return best;
}
static std::string plan_flags(const TunePlan &p)
{
    char buf[160];
    if (p.htsz_arg > 31 && (p.w & (p.w - 1))) snprintf(buf, sizeof buf, "-w %llu -buckets %u (extended table)", (unsigned long long)p.w, p.htsz_arg);
    else if (p.htsz_arg > 31) snprintf(buf, sizeof buf, "-w %.0f -buckets %u (extended table)", p.w_log2, p.htsz_arg);
    else snprintf(buf, sizeof buf, "-w %.0f -htsz %u%s", p.w_log2, p.htsz_arg, p.ext ? " -ext" : "");
    return buf;
}
static void tune(int gpu)
{
    bsgs_dev *dev = nullptr;
    if (bsgs_dev_open(gpu, &dev) != BSGS_OK) return;
    uint64_t fr = 0, tot = 0;
    int cus = 0;
    char name[256] = "";
    bsgs_dev_meminfo(dev, &fr, &tot); bsgs_dev_cu_count(dev, &cus); bsgs_dev_name(dev, name, sizeof name);
    const TuneAdvice a = tune_advice(fr);
    printf("GPU #%d %s: %d CUs, %.0f MB free -> suggested  -t 256 -b 256 -p 256 -w %.2f -htsz %u\n", gpu, name, cus, fr / 1048576.0, a.w_log2, a.htsz);
    if (a.ext) printf("GPU #%d extended table (w above the reference limit): -t 256 -b 256 -p 256 -w %u -htsz %u\n", gpu, a.ext_w_log2, a.ext_htsz);
    if (a.ext) printf("GPU #%d largest table for long searches (what -w auto takes for a range of 2^80 and more): -t 256 -b 256 -p 256 %s\n", gpu, plan_flags(tune_plan(fr, 120.0, 1, 1ull << 24)).c_str());
    bsgs_dev_close(dev);
}

// ---- checkpoint: saveCurentCNT 1_9_7File.pb:3897-3931 ------------------------------------------------------------
static std::string fingerprint(const Config &c)
{
    std::ostringstream s;
    s << c.t << c.b << c.p << c.w << c.pk << c.pke << (c.htsz_arg > 31 ? c.htsz_arg : c.htsz);     // Str(t)+Str(b)+Str(p)+Str(w)+pk+pke+Str(htsz)  (4635-4636); a bucket count stands for htsz
    return sha1_hex(s.str());
}
static void save_checkpoint(Shared &S)
{
    // the minimum counter over the GPUs' unfinished batches (a restart re-does at most the batches in flight); both locks are
    // held so that a batch cannot leave the dispenser between reading its counter and reading the in-flight table
    Scalar cnt;
    {
        std::lock_guard<std::mutex> lk(S.job_mutex);
        std::lock_guard<std::mutex> lk2(S.inflight_mutex);
        cnt = S.glob_key;
        for (size_t g = 0; g < S.inflight.size(); g++) if (S.inflight_valid[g] && hs::fe_cmp(S.inflight[g], cnt) < 0) cnt = S.inflight[g];
        if (S.joblog) { fprintf(S.joblog, "save %s\n", hs::fe_to_hex(cnt).c_str()); fflush(S.joblog); }
    }
    const std::string tmp = S.cfg.dir + "/currentwork.temp", dst = S.cfg.dir + "/currentwork.txt";
    {
        std::ofstream f(tmp, std::ios::binary);
        f << S.listpos << "\r\n" << S.mainpub_hex << "\r\n" << hs::fe_to_hex(cnt) << "\r\n" << fingerprint(S.cfg) << "\r\n";
    }
    rename(tmp.c_str(), dst.c_str());
}

// ---- -cpugen: the table and giants files built on the HOST CPU (BASELINE config 1 as it is worded; the reference's CPU-only generator is a program of its own,
// onlygen1_9_6File.pb:2915-3204, over lib/Curve64.pb).  Plumbing, not a fast path: k*G for k = 1..w by affine additions with batched normalisation (host_secp.h), one
// range of k per host thread; entries filed by bucket (counting sort), each bucket ascending by (hash, position) -- the order of the reference's sorted buckets
// (1_9_7File.pb:2771-2820) and of the GPU builder; images as in SURVEY.md Appendix C (1_9_7File.pb:3232-3444).  Byte-identical to the GPU builder's files (CPU test).
static void cpu_build_tables(uint64_t w, uint32_t htsz, uint8_t *htgpu, uint8_t *htcpu)
{
    const uint64_t items = 1ull << htsz;
    std::vector<uint64_t> key(w);                                     // low 64 bits of x(k*G) at index k - 1
    const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max(1u, std::thread::hardware_concurrency()), 64ull, (w + 65535) / 65536}));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([&, q]() {
        const uint64_t lo = w * q / nth, hi = w * (q + 1) / nth;      // k - 1 in [lo, hi)
        for (uint64_t first = lo; first < hi; first += 65536) {
            const size_t cnt = (size_t)std::min<uint64_t>(65536, hi - first);
            const std::vector<Affine> pts = hs::strided_multiples(hs::G, first + 1, 1, cnt);
            for (size_t i = 0; i < cnt; i++) key[first + i] = pts[i].x.l[0];
        }
    });
    for (auto &t : th) t.join();
    std::vector<uint32_t> off(items + 1, 0);
    for (uint64_t i = 0; i < w; i++) off[((uint32_t)key[i] & (uint32_t)(items - 1)) + 1]++;
    for (uint64_t b = 0; b < items; b++) off[b + 1] += off[b];       // off[b] = entries in buckets below b
    std::vector<uint64_t> ent(w);                                     // hash << 32 | position: ascending = (hash, position)
    {
        std::vector<uint32_t> cur(off.begin(), off.end() - 1);
        for (uint64_t i = 0; i < w; i++) ent[cur[(uint32_t)key[i] & (uint32_t)(items - 1)]++] = (key[i] >> 32 << 32) | i;
    }
    for (uint64_t b = 0; b < items; b++) std::sort(ent.begin() + off[b], ent.begin() + off[b + 1]);
    uint32_t *g = (uint32_t *)htgpu, *c = (uint32_t *)htcpu;
    memcpy(g, off.data(), 4 * (items + 1));                           // starts, then the total (= w)
    memcpy(c, off.data(), 4 * (items + 1));
    for (uint64_t i = 0; i < w; i++) {
        g[items + 1 + i] = (uint32_t)(ent[i] >> 32);
        c[items + 1 + 2 * i] = (uint32_t)(ent[i] >> 32);
        c[items + 1 + 2 * i + 1] = (uint32_t)ent[i];
    }
}
// G2[i] = (i + 1) * A, i < t*b*p, in the strided file layout (1_9_7File.pb:1831-1903, 1954-1970): the k-th MOST significant 32-bit word of coordinate c of G2[i]
// at u32 index c*8*maxnonce + ((i % p)*8 + k)*T + i / p, T = t*b
static void cpu_build_g2(const Affine &A, uint32_t t, uint32_t b, uint32_t p, uint8_t *g2)
{
    const uint64_t T = (uint64_t)t * b, maxnonce = T * p;
    uint32_t *out = (uint32_t *)g2;
    const unsigned nth = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t)std::max(1u, std::thread::hardware_concurrency()), 64ull, (maxnonce + 65535) / 65536}));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([&, q]() {
        const uint64_t lo = maxnonce * q / nth, hi = maxnonce * (q + 1) / nth;
        for (uint64_t first = lo; first < hi; first += 65536) {
            const size_t cnt = (size_t)std::min<uint64_t>(65536, hi - first);
            const std::vector<Affine> pts = hs::strided_multiples(A, first + 1, 1, cnt);
            for (size_t j = 0; j < cnt; j++) {
                const uint64_t i = first + j;
                for (int c = 0; c < 2; c++) {
                    const hs::Fe &v = c ? pts[j].y : pts[j].x;
                    for (int k = 0; k < 8; k++) out[(uint64_t)c * 8 * maxnonce + ((i % p) * 8 + k) * T + i / p] = (uint32_t)(v.l[3 - k / 2] >> (32 * (1 - k % 2)));
                }
            }
        }
    });
    for (auto &x : th) x.join();
}

// ---- the reference's limits on -w / -htsz for tables in ITS format (1_9_7File.pb:4412-4472): -w below 3069485951, -htsz below 32, and the "UNSAFE mode" question
// (answer Y on stdin to go on) where duplicate 32-bit values in one bucket become likely; then its warning about a -htsz that is too low.  Extended tables
// (built in GPU memory, no HT files: -ext, -w above 2^32, -buckets) are outside that format and outside these limits.  Returns "" to go on, else the exit message.
static std::string table_limits(uint64_t w, uint32_t htsz, FILE *answers)
{
    if (w >= 3069485951ull) return "-w should be less or equil to 3069485951 Or 2^31.515349920643907";
    if (htsz > 31) return "-htsz should be less than 32";
    static const struct { uint32_t htsz; uint64_t limit; const char *shown; } unsafe[] = {
        {27, 1331331443ull, "1331331443 or 2^30.310222637591963"}, {28, 1777178603ull, "1777178603 Or 2^30.726941530690112"}, {29, 3069485951ull, "3069485950 Or 2^31.515349920643907"},
        {30, 3069485951ull, "3069485951 Or 2^31.515349920643907"}, {31, 3069485951ull, "3069485951 Or 2^31.515349920643907"}};
    for (const auto &u : unsafe)
        if (htsz == u.htsz && w > u.limit) {
            printf("With -htsz %u value -w should be less or equil to %s\nDue to the possibility of duplicate values in the hash table\n"
                   "It is unsafe to use values higher than those specified above\nTo continue in UNSAFE mode type Y and press ENTER\n", u.htsz, u.shown);
            fflush(stdout);
            char line[64] = {0};
            if (!answers || !fgets(line, sizeof line, answers)) return " ";
            std::string ans(line);
            while (!ans.empty() && (ans.back() == '\n' || ans.back() == '\r')) ans.pop_back();
            if (ans != "Y") return " ";
        }
    const int need = (int)std::floor(std::log2((double)w)) - (int)htsz;
    if (need > 3) printf("WARNING! -htsz parametr is to low, should be at least %d\n", (int)std::floor(std::log2((double)w)) - 2);
    return "";
}

// ---- -selftest: the host-side logic that needs no GPU (CPU test tier, tests/test_host_logic.py) ------------------------
// prints "key value" lines: SHA1, the configuration fingerprint, host EC arithmetic, public-key parsing, the dispenser
// sequence and the table-free resolver, each for the inputs given on the command line
static int selftest(int argc, char **argv)
{
    std::vector<std::string> a(argv + 2, argv + argc);
    auto pt = [](const Affine &q) { return q.inf ? std::string("inf") : hs::fe_to_hex(q.x) + " " + hs::fe_to_hex(q.y); };
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i] == "sha1" && i + 1 < a.size()) printf("sha1 %s\n", sha1_hex(a[++i]).c_str());
        else if (a[i] == "fingerprint") {
            Config c; c.t = 256; c.b = 88; c.p = 130; c.w = 982162051; c.pk = "8000000000000000"; c.pke = "ffffffffffffffff"; c.htsz = 28;
            printf("fingerprint %s\n", fingerprint(c).c_str());
        } else if (a[i] == "mul" && i + 1 < a.size()) {
            Scalar k; if (!hs::fe_from_hex(k, a[++i])) return 2;
            printf("mul %s\n", pt(hs::point_mul(hs::G, k)).c_str());
        } else if (a[i] == "parse" && i + 1 < a.size()) {
            Affine q; const bool ok = hs::parse_pubkey(q, cut_hex(a[++i])) && hs::on_curve(q);
            printf("parse %s %s\n", ok ? pt(q).c_str() : "invalid", ok ? hs::compress_pubkey(q).c_str() : "");
        } else if (a[i] == "multiples" && i + 2 < a.size()) {                 // n multiples of k*G through the batched normalisation
            Scalar k; if (!hs::fe_from_hex(k, a[++i])) return 2;
            const size_t n = (size_t)atoi(a[++i].c_str());
            const std::vector<Affine> v = hs::multiples(hs::point_mul(hs::G, k), n);
            printf("multiples %s\n", pt(v.back()).c_str());
        } else if (a[i] == "jobs" && i + 5 < a.size()) {                     // dispenser: t b p w n -> counters and centres of n tiles
            Shared S;
            S.cfg.t = (uint32_t)atoi(a[i + 1].c_str()); S.cfg.b = (uint32_t)atoi(a[i + 2].c_str()); S.cfg.p = (uint32_t)atoi(a[i + 3].c_str());
            S.cfg.w = strtoull(a[i + 4].c_str(), nullptr, 10);
            const size_t n = (size_t)atoi(a[i + 5].c_str());
            Affine pub; if (!hs::parse_pubkey(pub, cut_hex(a[i + 6])) ) return 2;
            i += 6;
            S.maxnonce = (uint64_t)S.cfg.t * S.cfg.b * S.cfg.p;
            S.center_big = hs::sc_from_u128((hs::u128)S.cfg.p * S.cfg.w);
            S.center = hs::affine_neg(hs::point_mul(hs::G, S.center_big));
            S.gstep = hs::sc_mul_small(hs::sc_from_u128((hs::u128)S.maxnonce * S.cfg.w), 4);
            S.pubadd = hs::affine_neg(hs::point_mul(hs::G, S.gstep));
            S.glob_key = hs::fe_from_u64(1); S.glob_index = 0;
            S.walk_p0 = hs::point_add(hs::point_add(pub, hs::affine_neg(hs::point_mul(hs::G, S.glob_key))), S.center);
            std::vector<Tile> tiles;
            get_jobs(S, n, tiles);
            for (const Tile &t : tiles) printf("job %s %s\n", hs::fe_to_hex(t.key).c_str(), pt(tile_centre(S, t.index)).c_str());
        } else if (a[i] == "minibsgs" && i + 2 < a.size()) {                 // w (decimal), then hex scalars m: all b' <= w with x(b'G) = x(mG)
            const uint64_t w = strtoull(a[++i].c_str(), nullptr, 10);
            MiniBsgs mb; mb.build(w, 4);
            printf("minibsgs_bits %u\n", mb.mb);
            for (++i; i < a.size(); i++) {
                Scalar m; if (!hs::fe_from_hex(m, a[i])) return 2;
                std::string out;
                for (uint64_t b : mb.find(hs::point_mul(hs::G, m), w)) out += " " + std::to_string(b);
                printf("find %s%s\n", a[i].c_str(), out.c_str());
            }
        } else if (a[i] == "tune" && i + 1 < a.size()) {                      // free bytes -> the MI355X sizing advice (replaces Tune, 1_9_7File.pb:324-431)
            const TuneAdvice t = tune_advice(strtoull(a[++i].c_str(), nullptr, 10));
            printf("tune -w %.2f -htsz %u ext %d -w %u -htsz %u\n", t.w_log2, t.htsz, t.ext ? 1 : 0, t.ext_w_log2, t.ext_htsz);
        } else if (a[i] == "plan" && i + 3 < a.size()) {                      // free bytes, range bits, GPUs -> the table Tune picks for that range
            const uint64_t fr = strtoull(a[i + 1].c_str(), nullptr, 10);
            const TunePlan pl = tune_plan(fr, atof(a[i + 2].c_str()), atoi(a[i + 3].c_str()), 1ull << 24);
            i += 3;
            printf("plan %s | w %.2f htsz %u ext %d build %.3f search %.3f total %.3f\n", plan_flags(pl).c_str(), pl.w_log2, pl.htsz_arg, pl.ext ? 1 : 0, pl.build_s, pl.search_s, pl.total_s);
        } else if (a[i] == "htlookup" && i + 3 < a.size()) {                 // htCPU file, htsz, then hex 64-bit keys: positions found in RAM and by the two reads of -sf 1
            const std::string path = a[i + 1];
            const uint64_t items = 1ull << atoi(a[i + 2].c_str());
            struct stat st; if (stat(path.c_str(), &st) != 0) return 2;
            HostBuf img; if (!read_file(path, img, (uint64_t)st.st_size)) return 2;
            const int fd = open(path.c_str(), O_RDONLY); if (fd < 0) return 2;
            for (i += 3; i < a.size(); i++) {
                const uint64_t k = strtoull(a[i].c_str(), nullptr, 16);
                uint32_t p1[64], p2[64];
                const int n1 = htcpu_lookup(img, items, k, p1, 64), n2 = htcpu_lookup_file(fd, items, k, p2, 64);
                std::string o1, o2;
                for (int q = 0; q < std::min(n1, 64); q++) o1 += " " + std::to_string(p1[q]);
                for (int q = 0; q < std::min(n2, 64); q++) o2 += " " + std::to_string(p2[q]);
                printf("htlookup %s ram%s | file%s\n", a[i].c_str(), o1.c_str(), o2.c_str());
            }
            close(fd);
        } else if (a[i] == "limits" && i + 2 < a.size()) {                    // w (decimal), htsz: the reference's -w / -htsz limits and UNSAFE question (answer on stdin)
            const std::string m = table_limits(strtoull(a[i + 1].c_str(), nullptr, 10), (uint32_t)atoi(a[i + 2].c_str()), stdin);
            i += 2;
            printf("limits %s\n", m.empty() ? "ok" : m == " " ? "exit" : m.c_str());
        } else if (a[i] == "checkpoint" && i + 1 < a.size()) {                // next counter, then in-flight counters ("-" = idle GPU): the saved one
            Shared S;
            if (!hs::fe_from_hex(S.glob_key, a[++i])) return 2;
            for (++i; i < a.size(); i++) {
                Scalar v = hs::fe_from_u64(0);
                const bool valid = a[i] != "-";
                if (valid && !hs::fe_from_hex(v, a[i])) return 2;
                S.inflight.push_back(v); S.inflight_valid.push_back(valid);
            }
            S.cfg.dir = "/tmp"; S.mainpub_hex = "selftest"; S.joblog = stdout;
            save_checkpoint(S);
        } else { fprintf(stderr, "selftest: unknown item %s\n", a[i].c_str()); return 2; }
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc >= 2 && std::string(argv[1]) == "-selftest") return selftest(argc, argv);
    printf("BSGS MI355X (drop-in for bsgscudaHT 1.9.7-file0) on %s\n", bsgs_version());
    Shared S;
    Tables tables;
    S.tab = &tables;
    S.cfg = parse_args(argc, argv);
    const Config &c = S.cfg;
    // "[startup] <stage> <seconds>" lines: where the time before the first tile goes (bench.py's cold_time_to_solve_s reads them)
    auto t_stage = std::chrono::steady_clock::now();
    auto stage = [&](const char *what) {
        const auto n = std::chrono::steady_clock::now();
        printf("[startup] %-44s %.3fs\n", what, std::chrono::duration<double>(n - t_stage).count());
        t_stage = n;
    };
    const bool cpu_only = c.cpugen && c.onlygen;                      // the reference's CPU-only generator: no GPU is looked for, none is needed
    if (c.cpugen && (c.ext || c.w_auto)) die("-cpugen builds files in the reference`s format: not with -ext / -w auto / -w above 3069485950");
    int ngpu = 0;
    std::vector<int> gpus;
    if (!cpu_only) {
        CK(bsgs_dev_count(&ngpu));
        if (ngpu <= 0) die("No GPU found");
        if (c.devices.empty()) for (int i = 0; i < ngpu; i++) gpus.push_back(i);
        else { std::stringstream ss(c.devices); std::string tok; while (std::getline(ss, tok, ',')) gpus.push_back(atoi(tok.c_str())); }
    }
    stage("runtime + device discovery");
    for (int g : gpus) tune(g);
    stage("Tune lines (open / close every GPU)");
    if (!c.ext && !c.w_auto) { const std::string m = table_limits(c.w, c.htsz_arg, stdin); if (!m.empty()) die(m == " " ? "" : m); }
    // ---- range (1_9_7File.pb:4887-4943)
    if (!hs::fe_from_hex(S.start, c.pk) || hs::fe_is_zero(S.start)) die("Start range can`t be zero");
    printf("START RANGE= %s\n", hs::fe_to_hex(S.start).c_str());
    {   // the end of range is ALWAYS in force: privkeyend defaults to 1ffffffffffffffff and endrangeflag is set whenever it is
        // non-zero (1_9_7File.pb:210, 4897-4936); a key outside [pk, pke] ends with "Reached end of space"
        Scalar e;
        if (!hs::fe_from_hex(e, c.pke)) die("Invalid range (-pkend) length!!!");
        if (!hs::fe_is_zero(e)) {
            if (hs::fe_cmp(e, S.start) <= 0) die(c.pke_given ? "End range should be more than begin range!" : "End range must be more then start range");
            S.width = hs::sc_sub(e, S.start); S.end_range = true;
            int bits = 0;
            for (int l = 3; l >= 0 && !bits; l--) if (S.width.l[l]) bits = 64 * l + 64 - __builtin_clzll(S.width.l[l]);
            printf("  END RANGE= %s\nWIDTH RANGE= %s = 2^%d\n", hs::fe_to_hex(e).c_str(), hs::fe_to_hex(S.width).c_str(), bits);
        }
    }
    S.start_neg = hs::affine_neg(hs::point_mul(hs::G, S.start));
    int range_bits = 0;
    if (S.end_range) for (int l = 3; l >= 0 && !range_bits; l--) if (S.width.l[l]) range_bits = 64 * l + 64 - __builtin_clzll(S.width.l[l]);
    if (range_bits && !cpu_only) {
        // Tune for THIS range (the reference's Tune, 1_9_7File.pb:324-431, knows the GPU only): the table that minimises build + worst-case search
        bsgs_dev *dt = nullptr;
        uint64_t fr = 0, tot = 0;
        if (bsgs_dev_open(gpus[0], &dt) == BSGS_OK) { bsgs_dev_meminfo(dt, &fr, &tot); bsgs_dev_close(dt); }
        const TunePlan pl = tune_plan(fr, (double)range_bits, (int)gpus.size(), (uint64_t)c.t * c.b * c.p);
        printf("Tune for this range (2^%d keys, %zu GPU engine(s)): %s  -> table %.2fs + search at most %.2fs\n", range_bits, gpus.size(), plan_flags(pl).c_str(), pl.build_s, pl.search_s);
        if (c.w_auto) {
            Config &cw = S.cfg;
            cw.w = pl.w; cw.ext = pl.ext; cw.htsz_arg = pl.htsz_arg;
            cw.htsz = pl.htsz_arg <= 31 ? pl.htsz_arg : (uint32_t)std::floor(std::log2((double)pl.htsz_arg));
            printf("-w auto: Items number set to 2^%.2f=%llu, %s\n", pl.w_log2, (unsigned long long)cw.w, pl.ext ? "extended table in GPU memory (no HT files)" : "reference-format HT files");
        }
    } else if (c.w_auto) die("-w auto needs a range (-pk / -pke)");
    S.maxnonce = (uint64_t)c.t * c.b * c.p;
    // constants (1_9_7File.pb:4689-4712, 4759-4765)
    const Scalar two_w = hs::sc_from_u128((hs::u128)c.w * 2);
    S.addpubg = hs::affine_neg(hs::point_mul(hs::G, two_w));
    printf("GiantSUBvalue:%s\nGiantSUBpubkey: %s\n", hs::fe_to_hex(two_w).c_str(), hs::compress_pubkey(S.addpubg).c_str());
    S.center_big = hs::sc_from_u128((hs::u128)c.p * c.w);
    S.center = hs::affine_neg(hs::point_mul(hs::G, S.center_big));
    S.gstep = hs::sc_mul_small(hs::sc_from_u128((hs::u128)S.maxnonce * c.w), 4);
    S.pubadd = hs::affine_neg(hs::point_mul(hs::G, S.gstep));
    printf("Gstep: %s\n", hs::fe_to_hex(S.gstep).c_str());

    // ---- table files (Save_HTpacked 3645-3759, Save_Load_Giants 1905-2058): load, or build on the GPU and save
    const uint64_t ht_items = 1ull << c.htsz;
    const std::string gxhex = hs::fe_to_hex(hs::G.x);
    const std::string stem = c.dir + "/" + gxhex + "_" + std::to_string(c.w) + "_" + std::to_string(ht_items);
    const std::string f_gpu = stem + "_htGPUv0.BIN", f_cpu = stem + "_htCPUv0.BIN";
    const std::string f_g2 = c.dir + "/" + std::to_string(c.t) + "_" + std::to_string(c.b) + "_" + std::to_string(c.p) + "_" + std::to_string(c.w) + "_g2.BIN";
    HostBuf htgpu, g2;
    // files that were just generated are written by background threads while the start-up goes on (upload, bucket lines, scratch): the buffers they read
    // stay alive until `flush_writers` -- before the staging copies are released, and before any return
    std::vector<std::thread> writers;
    std::vector<std::function<void()>> pending_writes;               // started once the engines hold their tables: 15 GB going into the page cache next to the upload of the same
                                                                      // buffers slowed that upload from 0.25 s to 2 s (profiles/r07t_*)
    std::string saved_msg;
    auto start_writers = [&]() { for (auto &f : pending_writes) writers.emplace_back(f); pending_writes.clear(); };
    auto flush_writers = [&]() { start_writers(); for (auto &w : writers) w.join(); writers.clear(); if (!saved_msg.empty()) { fputs(saved_msg.c_str(), stdout); saved_msg.clear(); } };
    const uint64_t gpu_bytes = 4 * (ht_items + 1) + 4 * c.w, cpu_bytes = 4 * (ht_items + 1) + 8 * c.w, g2_bytes = 64 * S.maxnonce;
    bsgs_dev *d0 = nullptr;
    auto dev0 = [&]() { if (!d0) CK(bsgs_dev_open(gpus[0], &d0)); return d0; };
    if (c.ext) printf("Extended table: %llu items, built in GPU memory at start-up (no HT files)\n", (unsigned long long)c.w);
    else if (c.file_search && file_has_size(f_cpu, cpu_bytes) && read_file(f_gpu, htgpu, gpu_bytes) && (tables.htcpu_fd = open(f_cpu.c_str(), O_RDONLY)) >= 0)
        printf("Both HT files exist\nhtCPU is searched in its file (%.1f GB not loaded)\n", cpu_bytes / 1e9);
    else if (read_file(f_gpu, htgpu, gpu_bytes) && read_file(f_cpu, tables.htcpu, cpu_bytes)) printf("Both HT files exist\n");
    else {
        printf("Generate HT with %llu items on the %s\n", (unsigned long long)c.w, c.cpugen ? "host CPU" : "GPU");
        const auto t0 = std::chrono::steady_clock::now();
        htgpu.resize(gpu_bytes); tables.htcpu.resize(cpu_bytes);
        if (c.cpugen) cpu_build_tables(c.w, c.htsz, htgpu.data(), tables.htcpu.data());
        else CK(bsgs_build_baby_tables(dev0(), c.w, c.htsz, htgpu.data(), tables.htcpu.data(), BSGS_NO_INSTALL));
        pending_writes.emplace_back([&]() { write_file(f_cpu, tables.htcpu.data(), cpu_bytes); });
        pending_writes.emplace_back([&]() { write_file(f_gpu, htgpu.data(), gpu_bytes); });
        printf("Done in %.1fs\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    if (read_file(f_g2, g2, g2_bytes)) printf("Load BIN file:%s\n", f_g2.c_str());
    else {
        printf("Generate Giants Buffer: %llu items\n", (unsigned long long)S.maxnonce);
        g2.resize(g2_bytes);
        if (c.cpugen) cpu_build_g2(S.addpubg, c.t, c.b, c.p, g2.data());
        else {
            uint8_t axy[64];
            hs::affine_to_le(S.addpubg, axy, axy + 32);
            CK(bsgs_generate_g2(dev0(), axy, c.t, c.b, c.p));
            CK(bsgs_download_g2(dev0(), g2.data(), g2_bytes));
        }
        pending_writes.emplace_back([&]() { write_file(f_g2, g2.data(), g2_bytes); });
        saved_msg = "Save BIN file:" + f_g2 + "\n";                  // printed once the file IS on disk (flush_writers)
    }
    if (d0) { bsgs_dev_close(d0); d0 = nullptr; }
    stage("table + giants files (load, or build + save)");
    if (c.onlygen) { flush_writers(); printf("onlygen: files ready\n"); return 0; }


    // ---- recovery (-wl, 1_9_7File.pb:4634-4686)
    bool recovery = false; int rec_pos = 0; std::string rec_pub, rec_cnt;
    if (!c.recovery_file.empty()) {
        std::ifstream f(c.recovery_file);
        std::string l1, l2, l3, l4;
        auto strip = [](std::string s) { while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back(); return s; };
        if (!std::getline(f, l1) || !std::getline(f, l2) || !std::getline(f, l3) || !std::getline(f, l4)) die("Can`t read recovery file");
        if (strip(l4) != fingerprint(c)) die("Recovery file was made with other settings");
        rec_pos = atoi(strip(l1).c_str()); rec_pub = strip(l2); rec_cnt = strip(l3); recovery = true;
        printf("Recovery: listpos %d counter %s\n", rec_pos, rec_cnt.c_str());
    } else remove((c.dir + "/win.txt").c_str());                      // 1_9_7File.pb:4959-4963

    // ---- public keys (-pb or -infile, one per line, searched sequentially: 4370-4385, 4995-5168)
    std::vector<std::string> pubs;
    if (!c.infile.empty()) {
        std::ifstream f(c.infile);
        if (!f) die("Can`t open " + c.infile);
        std::string line;
        while (std::getline(f, line)) { while (!line.empty() && isspace((unsigned char)line.back())) line.pop_back(); if (!line.empty()) pubs.push_back(cut_hex(line)); }
    } else pubs.push_back(c.pub);

    std::thread mini_builder;                                           // extended tables: the resolver's own multiples of G, built on the host BEHIND the GPU start-up
    if (c.ext) mini_builder = std::thread([&tables, &c]() {
        const auto t0 = std::chrono::steady_clock::now();
        tables.mini.build(c.w, std::max(1u, std::thread::hardware_concurrency() / 2));
        printf("Resolver table: 2^%u multiples of G in %.1fs (behind the start-up)\n", tables.mini.mb, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    });
    // ---- how long a job is, in tiles (width / gstep): a job that is only a launch or two long (BASELINE config 4: a 64-bit range at -w 30 is 129 tiles) is dealt in small
    // batches that wait for their checker -- the reference, one tile per launch, stops at the hit (1_9_7File.pb:2442-2523); a full launch would always run to its end --,
    // its engines take scratch for such batches only, and with several keys to search two jobs run side by side (lanes)
    double job_tiles = 0.0;
    {
        auto as_double = [](const Scalar &v) { double r = 0.0; for (int l = 3; l >= 0; l--) r = r * 18446744073709551616.0 + (double)v.l[l]; return r; };
        job_tiles = S.end_range ? as_double(S.width) / as_double(S.gstep) + 1.0 : 0.0;
    }
    const double tpl_est = std::min(1024.0, std::max(48.0, (double)(192ull << 24) / (double)S.maxnonce));       // the engine's launch size at this geometry, memory permitting
    const bool short_job = job_tiles > 0.0 && job_tiles < 4.0 * tpl_est * (double)gpus.size();
    const size_t todo = pubs.size() - (recovery && rec_pos >= 1 && (size_t)rec_pos <= pubs.size() ? (size_t)rec_pos - 1 : 0);
    size_t lanes = 1;
    if (c.lanes > 0) lanes = (size_t)c.lanes;
    else if (short_job && todo >= 4 && !c.ext && c.joblog.empty()) lanes = 2;
    lanes = std::max<size_t>(1, std::min(lanes, todo));
    if (short_job) {
        // about six batches per GPU and job (fourteen with two lanes: the other lane's launch hides this one's boundaries), not below 16 (8) tiles: the narrow batchings keep
        // small launches at 35-38 G, and with the key anywhere in the range 0.55-0.6 of the tiles are searched on average instead of all of them
        const double per_job = getenv("BSGS_SHORT_JOB_BATCHES") ? std::max(1.0, atof(getenv("BSGS_SHORT_JOB_BATCHES"))) : (lanes > 1 ? 14.0 : 6.0);      // (the variable: A-B runs; 1000 keys of config 4: 10 -> 64-66 s, 14 -> 61.8 s, profiles/r07g_*)
        S.batch_hint = (uint32_t)std::min(tpl_est, std::max(lanes > 1 ? 8.0 : 16.0, std::ceil(job_tiles / (per_job * (double)gpus.size()))));
        printf("Short jobs (%.0f tiles each): dealt in batches of %u tiles%s\n", job_tiles, S.batch_hint, lanes > 1 ? ", two public keys searched side by side (an engine each per GPU)" : "");
    }
    if (lanes > 1) { const std::vector<int> base = gpus; for (size_t l = 1; l < lanes; l++) gpus.insert(gpus.end(), base.begin(), base.end()); }
    std::vector<bsgs_dev *> devs(gpus.size(), nullptr);
    {
        for (size_t gi = 0; gi < gpus.size(); gi++) devs[gi] = open_dev(gpus[gi]);
        if (S.batch_hint) for (bsgs_dev *d : devs) CK(bsgs_set_tiles_per_launch(d, S.batch_hint));     // scratch (and its placement) for the batches this run will launch, not for 192 tiles
        // the engines of lane 0 are loaded (and, several GPUs, compared); the engines of the other lanes are TWINS of theirs on the same GPU: they probe the same table in
        // place (bsgs_share_tables) -- no second 21 GiB to place, clear and copy at -w 30 -- with giants and chain scratch of their own
        const size_t primaries = gpus.size() / lanes;
        const std::vector<int> gpus0(gpus.begin(), gpus.begin() + (long)primaries);
        const std::vector<bsgs_dev *> devs0(devs.begin(), devs.begin() + (long)primaries);
        load_engines(S, gpus0, devs0, htgpu, g2);
        if (devs0.size() > 1 && c.verify_replicas) verify_replicas(gpus0, devs0);
        if (lanes > 1) {
            const auto t0 = std::chrono::steady_clock::now();
            for (size_t gi = primaries; gi < devs.size(); gi++) { CK(bsgs_share_tables(devs[gi % primaries], devs[gi])); CK(bsgs_prepare(devs[gi])); print_placement(gpus[gi], gi, devs[gi]); }
            printf("[startup] %-44s %.3fs\n", "twin engines of the other lanes (shared tables)", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (c.ref_quirks) { for (bsgs_dev *d : devs) CK(bsgs_set_flags(d, BSGS_FLAG_REFERENCE_QUIRKS)); printf("Reference-quirk mode: NEGMODP borrow bug reproduced\n"); }
    }
    stage("upload, bucket lines, chain scratch, replicas");
    start_writers();
    if (!c.joblog.empty()) { S.joblog = fopen(c.joblog.c_str(), "w"); if (!S.joblog) die("Can`t create " + c.joblog); }
    // freshly generated files keep being written BEHIND the search (their writers are joined before the process leaves; a file appears under its name only once it is
    // complete: write_file): the 13 GB of HT files of a -w 30 run cost the first jobs nothing.  Only the resolver's table must be there before the first hit.
    if (mini_builder.joinable()) mini_builder.join();
    stage("resolver table (behind the start-up)");

    // ---- the jobs: one public key after the other (1_9_7File.pb:4995-5168) -- or, when a job is only a launch or two long (BASELINE config 4: 1000 keys over a 64-bit
    // range), `lanes` of them side by side, each on an engine of its own per GPU: while one job waits for its checker, dispenses, or parses the next key, the other's
    // launch keeps the GPU busy, and no tile is searched on speculation.  win.txt and the console keep the list order; currentwork.txt describes the OLDEST job in flight.
    const size_t G = gpus.size() / lanes;                             // engines per lane
    std::vector<std::unique_ptr<Shared>> lane_state;
    for (size_t l = 0; l < lanes; l++) {
        std::unique_ptr<Shared> J(new Shared());
        J->cfg = S.cfg; J->maxnonce = S.maxnonce; J->center_big = S.center_big; J->gstep = S.gstep; J->start = S.start; J->width = S.width; J->end_range = S.end_range;
        J->addpubg = S.addpubg; J->center = S.center; J->pubadd = S.pubadd; J->start_neg = S.start_neg; J->tab = S.tab; J->joblog = l == 0 ? S.joblog : nullptr; J->batch_hint = S.batch_hint;
        lane_state.push_back(std::move(J));
    }
    struct JobOut { bool done = false, found = false; std::string text, win; };
    std::vector<JobOut> outs(pubs.size());
    std::mutex out_mutex;
    size_t next_emit = 0, next_job = 0;
    int finditems = 0;
    std::vector<int> lane_listpos(lanes, 0);                          // list position each lane works on (0 = idle): the checkpoint belongs to the smallest
    std::mutex lane_mutex;
    const bool live = lanes == 1;                                     // one lane: every line appears as it happens; several: a job's lines are printed when its turn in the list comes
    auto emit = [&]() {                                               // under out_mutex: print / append to win.txt everything that is complete, in list order
        while (next_emit < outs.size() && outs[next_emit].done) {
            JobOut &o = outs[next_emit];
            if (!live) fputs(o.text.c_str(), stdout);
            if (o.found) {
                std::ofstream f(c.dir + "/win.txt", std::ios::app | std::ios::binary);
                f << o.win;
                finditems++;
            }
            o.text.clear();
            next_emit++;
        }
        fflush(stdout);
    };
    bool tuned = false;
    auto run_lane = [&](size_t l) {
        Shared &J = *lane_state[l];
        const std::vector<int> lgpus(gpus.begin() + l * G, gpus.begin() + (l + 1) * G);
        const std::vector<bsgs_dev *> ldevs(devs.begin() + l * G, devs.begin() + (l + 1) * G);
        for (;;) {
            size_t li;
            bool resumed = false;
            {
                std::lock_guard<std::mutex> lk(lane_mutex);
                while (next_job < pubs.size() && recovery && (int)next_job + 1 != rec_pos) { { std::lock_guard<std::mutex> lo(out_mutex); outs[next_job].done = true; } next_job++; }      // -wl: everything before the saved position is skipped
                if (next_job >= pubs.size()) { lane_listpos[l] = 0; break; }
                li = next_job++;
                lane_listpos[l] = (int)li + 1;
                if (recovery) { resumed = true; recovery = false; }       // this is the saved position: it resumes from the saved counter, everything after it starts fresh
            }
            JobOut &o = outs[li];
            auto say = [&](const char *fmt, ...) {
                char buf[1024];
                va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
                if (live) { fputs(buf, stdout); fflush(stdout); } else o.text += buf;
            };
            J.listpos = (int)li + 1;
            if (!hs::parse_pubkey(J.realpub, pubs[li]) || !hs::on_curve(J.realpub)) die("Invalid Public Key (-pb) length!!!");
            J.mainpub_hex = hs::fe_to_hex(J.realpub.x) + hs::fe_to_hex(J.realpub.y);
            if (resumed && J.mainpub_hex != rec_pub) die("Find position but the keys are different");
            say("\nFindpubkey  : %s\n", hs::compress_pubkey(J.realpub).c_str());
            J.findpub = hs::point_add(J.realpub, J.start_neg);             // 1_9_7File.pb:5042
            say("Searchpubkey: %s\n", hs::compress_pubkey(J.findpub).c_str());
            // dispenser seed (1_9_7File.pb:5046-5064)
            J.glob_key = hs::fe_from_u64(1);
            if (resumed && !hs::fe_from_hex(J.glob_key, rec_cnt)) die("bad counter");
            J.glob_index = 0;
            J.walk_p0 = hs::point_add(hs::point_add(J.findpub, hs::affine_neg(hs::point_mul(hs::G, J.glob_key))), J.center);
            if (!c.host_centres) {
                if (J.walk_p0.inf) die("the public key equals (counter + p*w)*G: the first tile centre is the point at infinity");
                uint8_t p0[64], st[64];
                hs::affine_to_le(J.walk_p0, p0, p0 + 32); hs::affine_to_le(J.pubadd, st, st + 32);
                for (bsgs_dev *d : ldevs) CK(bsgs_set_walk(d, p0, st));      // from here on the host only advances the counter
                if (c.tune && !tuned && l == 0) {
                    // once per run: the launch time depends on which physical memory the driver handed out for the chain scratch and the
                    // bucket lines; try a few placements on every GPU (in parallel) and keep the fastest
                    tuned = true;
                    std::vector<std::array<float, 7>> res(ldevs.size());
                    std::vector<int> rcs(ldevs.size(), 0);
                    std::vector<std::string> why(ldevs.size());
                    std::vector<std::thread> tt;
                    for (size_t gi = 0; gi < ldevs.size(); gi++) tt.emplace_back([&, gi]() {
                        uint32_t kept[2] = {0, 0};
                        rcs[gi] = bsgs_tune_placement(ldevs[gi], 3, res[gi].data(), kept, &res[gi][6]);
                        if (rcs[gi]) why[gi] = bsgs_last_error();           // the error text is per thread
                    });
                    for (auto &t : tt) t.join();
                    for (size_t gi = 0; gi < ldevs.size(); gi++) {
                        if (rcs[gi]) { say("GPU #%d: placement tuning skipped (%s)\n", lgpus[gi], why[gi].c_str()); continue; }
                        say("GPU #%d: placement tuned, %.1f -> %.1f ms per launch\n", lgpus[gi], res[gi][0], res[gi][6]);
                    }
                }
            }
            J.past_end = false;
            J.job_tiles = job_tiles; J.ngpus = (int)G;
            J.quit = false; J.all_done = false; J.found = false; J.gpus_finished = 0; J.steps_done = 0; J.tiles_done = 0; J.hits_checked = 0; J.hits_pushed = 0; J.checker_ns = 0;
            const auto t0 = std::chrono::steady_clock::now();
            Scalar one = hs::fe_from_u64(1), two = hs::fe_from_u64(2);
            Scalar trivial;                                                 // keys 1 and 2 are answered without search (5069-5107)
            bool is_trivial = false;
            for (const Scalar &k : {one, two}) { const Affine q = hs::point_mul(hs::G, k); if (hs::fe_equal(q.x, J.realpub.x) && hs::fe_equal(q.y, J.realpub.y)) { trivial = k; is_trivial = true; } }
            if (!is_trivial) {
                const unsigned nchk = c.ext ? std::max(2u, std::min(16u, std::thread::hardware_concurrency() / 4)) : 1u;   // false positives cost a small BSGS each
                std::vector<std::thread> chk;
                for (unsigned q = 0; q < nchk; q++) chk.emplace_back(checker_thread, &J);
                std::vector<std::thread> th;
                J.inflight.assign(G, hs::fe_from_u64(0)); J.inflight_valid.assign(G, false);
                for (size_t gi = 0; gi < G; gi++) th.emplace_back(gpu_thread, &J, lgpus[gi], (int)gi, ldevs[gi]);
                auto last_save = std::chrono::steady_clock::now();
                uint64_t last_steps = 0; auto last_t = t0;
                while (J.gpus_finished.load() < (int)G) {
                    { std::unique_lock<std::mutex> lk(J.done_mutex); J.done_cv.wait_for(lk, std::chrono::milliseconds(200), [&] { return J.gpus_finished.load() >= (int)G; }); }
                    const auto now = std::chrono::steady_clock::now();
                    if (live && std::chrono::duration<double>(now - last_t).count() >= 2.0) {       // progress line 5119-5142
                        const uint64_t st = J.steps_done.load();
                        const double rate = (st - last_steps) / std::chrono::duration<double>(now - last_t).count();
                        Scalar cnt; { std::lock_guard<std::mutex> lk(J.job_mutex); cnt = J.glob_key; }
                        printf("\rCnt:%s [%d] = %.0f MKeys/s x2^%.2f=2^%.2f   ", hs::fe_to_hex(cnt).c_str() + 40, (int)G, rate / 1048576.0,
                               std::log2(2.0 * c.w), rate > 0 ? std::log2(rate * 2.0 * c.w) : 0.0);
                        fflush(stdout);
                        last_steps = st; last_t = now;
                    }
                    if (std::chrono::duration<double>(now - last_save).count() >= c.wt || J.joblog) {
                        bool oldest = true;                                 // currentwork.txt: the oldest job in flight (a restart re-does the younger ones from their start)
                        { std::lock_guard<std::mutex> lk(lane_mutex); for (int lp : lane_listpos) oldest &= lp == 0 || lp >= J.listpos; }
                        if (oldest) save_checkpoint(J);
                        last_save = now;
                    }
                }
                for (auto &x : th) x.join();
                // drain the checker queue, then stop it
                for (;;) { { std::lock_guard<std::mutex> lk(J.chk_mutex); if (J.checker.empty()) break; } if (J.quit.load()) break; std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
                J.all_done = true; J.chk_cv.notify_all();
                for (auto &x : chk) x.join();
            } else { J.winkey = trivial; J.found = true; }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (J.found) {                                                  // win.txt 1_9_7File.pb:5146-5160
                const std::string head = "KEY[" + std::to_string(J.listpos) + "]: ";
                const std::string l1 = head + "0x" + hs::fe_to_hex(J.winkey);
                const std::string l2 = std::string(head.size() - 5, ' ') + "Pub: " + hs::compress_pubkey(J.realpub);
                say("\n****************************\n%s\n%s\n****************************\n", l1.c_str(), l2.c_str());
                o.found = true; o.win = l1 + "\r\n" + l2 + "\r\n";
            } else say("\nReached end of space\n");
            say("Job time %.2fs, %llu tiles, %.3e giant steps\n", secs, (unsigned long long)J.tiles_done.load(), (double)J.steps_done.load());
            say("Checker: %llu hits resolved in %.3fs of CPU time (%.2f%% of one core)\n", (unsigned long long)J.hits_checked.load(), J.checker_ns.load() * 1e-9,
                secs > 0 ? 100.0 * J.checker_ns.load() * 1e-9 / secs : 0.0);
            { std::lock_guard<std::mutex> lk(out_mutex); o.done = true; emit(); }
        }
    };
    {
        std::vector<std::thread> lt;
        for (size_t l = 1; l < lanes; l++) lt.emplace_back(run_lane, l);
        run_lane(0);
        for (auto &t : lt) t.join();
        std::lock_guard<std::mutex> lk(out_mutex);
        emit();
    }
    if (S.joblog) fclose(S.joblog);
    flush_writers();                                                  // the files that were still being written behind the search
    htgpu.release();                                                  // host staging copies (1_9_7File.pb:4818-4843)
    g2.release();
    printf("Found %d of %zu\n", finditems, pubs.size());
    fflush(stdout);
    if (getenv("BSGS_HOST_CLEAN_EXIT")) { for (size_t gi = devs.size(); gi-- > 0;) bsgs_dev_close(devs[gi]); return 0; }      // twins (borrowed tables) before their owners
    // the search is over and every file is on disk: leave without the runtime's teardown (freeing a few hundred GiB of device memory buffer by buffer and unloading
    // the code objects costs 0.15-0.3 s of a 64-bit solve that takes one; the driver reclaims everything with the process)
    _exit(0);
}
