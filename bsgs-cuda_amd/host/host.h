// host.h -- declarations shared by the translation units of bsgs_mi355x, the C++ host of the MI355X BSGS solver (see bsgs_host.cpp for the reference
// file:line map).  host_config.cpp: command line, limits, checkpoint; host_files.cpp: table files and the CPU-only generator; host_resolver.cpp: dispenser and hit
// resolver; host_tune.cpp: Tune; host_engines.cpp: per-GPU engines (load, verify, search thread); host_selftest.cpp: -selftest; bsgs_host.cpp: main.
#pragma once
#include "../../include/bsgs_hip.h"
#include "../csrc/host_secp.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <set>
#include <sstream>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

using hs::Affine;
using hs::Scalar;

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
    bool verify_replicas = true;                   // verify before searching (-noverify skips): census + sampled k*G + sampled giants on every engine, and -- several engines -- table checksums and the hits of one tile compared across them
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

[[noreturn]] void die(const std::string &msg);
std::string cut_hex(std::string s);
std::string sha1_hex(const std::string &msg);
void usage(const Config &c);
Config parse_args(int argc, char **argv);
std::string table_limits(uint64_t w, uint32_t htsz, FILE *answers);
std::string fingerprint(const Config &c);

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
    void resize(uint64_t bytes);
    void release() { free(p); p = nullptr; n = 0; }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    uint64_t size() const { return n; }
    const uint8_t &operator[](uint64_t i) const { return p[i]; }
};
bool file_has_size(const std::string &path, uint64_t expect);
bool read_file(const std::string &path, HostBuf &out, uint64_t expect);
void write_file(const std::string &path, const void *p, uint64_t n);
void cpu_build_tables(uint64_t w, uint32_t htsz, uint8_t *htgpu, uint8_t *htcpu);
void cpu_build_g2(const Affine &A, uint32_t t, uint32_t b, uint32_t p, uint8_t *g2);

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

void save_checkpoint(Shared &S);
Affine tile_centre(const Shared &S, uint64_t index);
size_t get_jobs(Shared &S, size_t n, std::vector<Tile> &out, int slot = -1);
int htcpu_lookup_file(int fd, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max);
int htcpu_lookup(const HostBuf &img, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max);
void checker_thread(Shared *S);

struct TuneAdvice { double w_log2; uint32_t htsz; bool ext; uint32_t ext_w_log2, ext_htsz; };
struct TunePlan { double w_log2; uint32_t htsz_arg; bool ext; double build_s, search_s, total_s; uint64_t w; };      // w = the number of baby points itself (it need not be a power of two)
TuneAdvice tune_advice(uint64_t free_bytes);
TunePlan tune_plan(uint64_t free_bytes, double range_bits, int n_gpus, uint64_t maxnonce);
std::string plan_flags(const TunePlan &p);
void tune(int gpu);

bsgs_dev *open_dev(int gpu);
void print_placement(int gpu, size_t gi, bsgs_dev *dev);
void load_engines(const Shared &S, const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs, const HostBuf &htgpu, const HostBuf &g2);
void verify_replicas(const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs);
void verify_tables(const Shared &S, const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs);
void test_corrupt_engine(const std::vector<bsgs_dev *> &devs);       // test build only (-DBSGS_TEST_HOOKS); a no-op in the shipped host
void per_gpu(const std::vector<int> &gpus, const std::function<void(size_t)> &fn);
void gpu_thread(Shared *S, int gpu, int slot, bsgs_dev *dev);
int selftest(int argc, char **argv);
