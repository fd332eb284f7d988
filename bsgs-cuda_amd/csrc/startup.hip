// startup.hip -- start-up of N engines driven by ONE process (the C++ host: one thread per GPU, 1_9_7File.pb:4769-4843): how the replicas of the giants and of
// the baby table get onto every GPU.  The reference uploads one host buffer to every GPU over PCIe (1_9_7File.pb:2337, 2350).  Here:
//
//   transports   RCCL over xGMI (ncclCommInitAll: one communicator per engine, all in this process; ncclBroadcast / ncclAllGather inside one group) when the engines sit
//                on distinct GPUs and librccl can be loaded -- it is dlopen'ed on first use, so a one-GPU run never pays for it --, else direct peer copies
//                (hipMemcpyPeerAsync: the only way when one GPU is listed twice, `-d 0,0`, which RCCL refuses);
//   strategies   for extended tables (built on the GPU, w >= 2^32: nothing to upload): BROADCAST -- engine 0 builds, everybody else receives;  LOCAL -- every engine
//                builds its own replica, concurrently, no link traffic;  ALLGATHER -- every engine generates every point but files only the 1/N of the buckets it owns,
//                then the slices are all-gathered.  Expected seconds of each at N = 8: DESIGN.md 7.
// Whatever the route, the host compares the replicas afterwards (bsgs_table_checksum + one probe tile: bsgs_mi355x verify_replicas).
#include "bsgs_internal.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {
// ---- RCCL, loaded on demand ------------------------------------------------------------------------------------------------------------------
struct RcclApi {
    void *h = nullptr;
    std::string why;                            // why it is not available
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi &rccl_api()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *cands[] = {getenv("BSGS_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *c : cands) {
            if (!c || !*c) continue;
            api.h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
            if (api.h) break;
            api.why = dlerror();
        }
        if (!api.h) return;
        bool ok = true;
        auto sym = [&](auto &fn, const char *name) { fn = (std::remove_reference_t<decltype(fn)>)dlsym(api.h, name); if (!fn) { ok = false; api.why = std::string("librccl lacks ") + name; } };
        sym(api.GetVersion, "ncclGetVersion"); sym(api.CommInitAll, "ncclCommInitAll"); sym(api.CommDestroy, "ncclCommDestroy");
        sym(api.GroupStart, "ncclGroupStart"); sym(api.GroupEnd, "ncclGroupEnd"); sym(api.Broadcast, "ncclBroadcast"); sym(api.AllGather, "ncclAllGather");
        sym(api.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(api.h); api.h = nullptr; }
    });
    return api;
}
}  // namespace

// ---- the fabric between the engines of one process ---------------------------------------------------------------------------------------------
struct bsgs_fabric {
    std::vector<bsgs_dev *> devs;
    bool rccl = false;
    std::vector<ncclComm_t> comms;
    std::string name;
};

#define NCCLCHK(f, call)                                                                                                                   \
    do {                                                                                                                                  \
        ncclResult_t r_ = (call);                                                                                                         \
        if (r_ != ncclSuccess) return bsgs_fail(BSGS_ERR_HIP, "%s -> %s (%s:%d)", #call, rccl_api().GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

static int fabric_sync(bsgs_fabric *f)
{
    for (bsgs_dev *d : f->devs) { HIPCHK(hipSetDevice(d->id)); HIPCHK(hipStreamSynchronize(d->stream)); }
    return BSGS_OK;
}

int bsgs_fabric_open(bsgs_fabric **out, bsgs_dev *const *devs, int n, uint32_t transport)
{
    if (!out || !devs || n < 1) return fail(BSGS_ERR_ARG, "null");
    if (transport > BSGS_TRANSPORT_PEER) return fail(BSGS_ERR_ARG, "transport %u: 0 auto, 1 RCCL, 2 peer copies", transport);
    if (const char *e = getenv("BSGS_TRANSPORT")) {               // diagnostics: override the caller's choice
        if (!strcmp(e, "rccl")) transport = BSGS_TRANSPORT_RCCL;
        else if (!strcmp(e, "peer")) transport = BSGS_TRANSPORT_PEER;
    }
    bool distinct = true;
    for (int i = 0; i < n; i++) {
        if (!devs[i]) return fail(BSGS_ERR_ARG, "null device %d", i);
        for (int j = 0; j < i; j++) distinct &= devs[i]->id != devs[j]->id;
    }
    bsgs_fabric *f = new bsgs_fabric();
    f->devs.assign(devs, devs + n);
    bool want_rccl = transport == BSGS_TRANSPORT_RCCL || (transport == BSGS_TRANSPORT_AUTO && distinct && n > 1);
    if (want_rccl && !distinct) { delete f; return fail(BSGS_ERR_ARG, "RCCL needs distinct GPUs (one is listed twice): use peer copies"); }
    if (want_rccl) {
        RcclApi &R = rccl_api();
        if (!R.h) {
            if (transport == BSGS_TRANSPORT_RCCL) { delete f; return fail(BSGS_ERR_STATE, "librccl could not be loaded: %s", R.why.c_str()); }
            fprintf(stderr, "bsgs: librccl not available (%s): replicas travel by peer copies\n", R.why.c_str());
            want_rccl = false;
        }
    }
    if (want_rccl) {
        RcclApi &R = rccl_api();
        std::vector<int> ids(n);
        for (int i = 0; i < n; i++) ids[i] = devs[i]->id;
        f->comms.assign(n, nullptr);
        const ncclResult_t r = R.CommInitAll(f->comms.data(), n, ids.data());
        if (r != ncclSuccess) {
            const std::string why = R.GetErrorString(r);
            f->comms.clear();
            if (transport == BSGS_TRANSPORT_RCCL) { delete f; return fail(BSGS_ERR_HIP, "ncclCommInitAll over %d GPU(s): %s", n, why.c_str()); }
            fprintf(stderr, "bsgs: ncclCommInitAll failed (%s): replicas travel by peer copies\n", why.c_str());
        } else {
            int v = 0;
            (void)R.GetVersion(&v);
            f->rccl = true;
            f->name = "RCCL " + std::to_string(v / 10000) + "." + std::to_string(v / 100 % 100) + "." + std::to_string(v % 100) + " (ncclCommInitAll, " + std::to_string(n) + " rank(s) in one process)";
        }
    }
    if (!f->rccl) {
        // peer copies: every engine must be able to read every other GPU's memory (bucket lines composed of mapped chunks grant that when they are mapped: placement.hip)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                if (devs[i]->id == devs[j]->id) continue;
                int can = 0;
                if (hipSetDevice(devs[i]->id) != hipSuccess) continue;
                if (hipDeviceCanAccessPeer(&can, devs[i]->id, devs[j]->id) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(devs[j]->id, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { delete f; return fail(BSGS_ERR_HIP, "peer access %d -> %d: %s", devs[i]->id, devs[j]->id, hipGetErrorString(pe)); }
                    (void)hipGetLastError();
                }
            }
        f->name = "peer copies (hipMemcpyPeerAsync)";
    }
    *out = f;
    return BSGS_OK;
}
void bsgs_fabric_close(bsgs_fabric *f)
{
    if (!f) return;
    if (f->rccl) for (size_t i = 0; i < f->comms.size(); i++) if (f->comms[i]) { (void)hipSetDevice(f->devs[i]->id); (void)rccl_api().CommDestroy(f->comms[i]); }
    delete f;
}
const char *bsgs_fabric_name(const bsgs_fabric *f) { return f ? f->name.c_str() : ""; }
int bsgs_fabric_is_rccl(const bsgs_fabric *f) { return f && f->rccl ? 1 : 0; }

// one-to-all: bufs[i] = the buffer on engine i (bufs[root] holds the data); every engine's stream is drained on return
int bsgs_fabric_broadcast(bsgs_fabric *f, void *const *bufs, size_t bytes, int root)
{
    const int n = (int)f->devs.size();
    if (root < 0 || root >= n) return fail(BSGS_ERR_ARG, "root %d of %d", root, n);
    if (!bytes) return BSGS_OK;
    if (f->rccl) {
        RcclApi &R = rccl_api();
        const size_t piece = 4ull << 30;                         // one collective per 4 GiB: every call's count stays far from any 32-bit limit inside the library
        for (size_t off = 0; off < bytes; off += piece) {
            const size_t cnt = std::min(piece, bytes - off);
            // a group that was started is always ended, also when a call inside it fails (the first failure is the one reported)
            NCCLCHK(f, R.GroupStart());
            ncclResult_t bad = ncclSuccess;
            hipError_t hbad = hipSuccess;
            for (int i = 0; i < n && bad == ncclSuccess && hbad == hipSuccess; i++) {
                hbad = hipSetDevice(f->devs[i]->id);
                if (hbad == hipSuccess) bad = R.Broadcast((const char *)bufs[i] + off, (char *)bufs[i] + off, cnt, ncclUint8, root, f->comms[i], f->devs[i]->stream);
            }
            const ncclResult_t ge = R.GroupEnd();
            HIPCHK(hbad);
            NCCLCHK(f, bad);
            NCCLCHK(f, ge);
        }
    } else {
        for (int i = 0; i < n; i++) {
            if (i == root || bufs[i] == bufs[root]) continue;
            HIPCHK(hipSetDevice(f->devs[i]->id));
            HIPCHK(hipMemcpyPeerAsync(bufs[i], f->devs[i]->id, bufs[root], f->devs[root]->id, bytes, f->devs[i]->stream));
        }
    }
    return fabric_sync(f);
}
// all-gather in place: bufs[i] = N * slice_bytes on engine i, its own slice (number i) already there
int bsgs_fabric_allgather(bsgs_fabric *f, void *const *bufs, size_t slice_bytes)
{
    const int n = (int)f->devs.size();
    if (!slice_bytes || (n == 1 && !f->rccl)) return fabric_sync(f);
    if (f->rccl) {                                                // (one rank: the collective is still issued -- in place, nothing moves -- so that a one-GPU lease exercises the call)
        RcclApi &R = rccl_api();
        // counted in 64-bit words where the slice allows (line slices always do): the element count of a 24 GiB slice (-w 35 over 8 GPUs) stays below 2^32
        const bool wide = slice_bytes % 8 == 0;
        NCCLCHK(f, R.GroupStart());
        ncclResult_t bad = ncclSuccess;
        hipError_t hbad = hipSuccess;
        for (int i = 0; i < n && bad == ncclSuccess && hbad == hipSuccess; i++) {
            hbad = hipSetDevice(f->devs[i]->id);
            if (hbad == hipSuccess)
                bad = R.AllGather((const char *)bufs[i] + (size_t)i * slice_bytes, bufs[i], wide ? slice_bytes / 8 : slice_bytes, wide ? ncclUint64 : ncclUint8, f->comms[i], f->devs[i]->stream);
        }
        const ncclResult_t ge = R.GroupEnd();
        HIPCHK(hbad);
        NCCLCHK(f, bad);
        NCCLCHK(f, ge);
    } else {
        for (int i = 0; i < n; i++) {
            HIPCHK(hipSetDevice(f->devs[i]->id));
            for (int k = 1; k < n; k++) {                        // engine i starts with its right-hand neighbour's slice: at any moment every source is read by one engine
                const int j = (i + k) % n;
                HIPCHK(hipMemcpyPeerAsync((char *)bufs[i] + (size_t)j * slice_bytes, f->devs[i]->id, (const char *)bufs[j] + (size_t)j * slice_bytes, f->devs[j]->id, slice_bytes,
                                          f->devs[i]->stream));
            }
        }
    }
    return fabric_sync(f);
}

// ---- extended tables for N engines ----------------------------------------------------------------------------------------------------------------
namespace {
double since(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
// fn(i) for every engine, one host thread per GPU (the library's entry points select their device themselves): engines on distinct GPUs run concurrently, engines that
// share a GPU (`-d 0,0`) one after the other -- a table build takes the free memory of its GPU for itself while it runs.  The first failure's code and text are returned.
int parallel(bsgs_dev *const *devs, int n, const std::function<int(int)> &fn)
{
    std::vector<int> rc(n, BSGS_OK);
    std::vector<std::string> why(n);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++) {
        bool first = true;
        for (int j = 0; j < i; j++) first &= devs[j]->id != devs[i]->id;
        if (!first) continue;
        th.emplace_back([&, i] {
            for (int k = i; k < n; k++) {
                if (devs[k]->id != devs[i]->id) continue;
                rc[k] = fn(k);
                if (rc[k]) why[k] = bsgs_last_error();
            }
        });
    }
    for (auto &t : th) t.join();
    for (int i = 0; i < n; i++) if (rc[i]) return bsgs_fail(rc[i], "engine %d: %s", i, why[i].c_str());
    return BSGS_OK;
}
}  // namespace

extern "C" int bsgs_startup_ext_tables(bsgs_dev *const *devs, int n, uint64_t w, uint32_t htsz, uint32_t layout, uint32_t strategy, uint32_t transport,
                                       bsgs_startup_report *rep)
{
    if (!devs || n < 1) return fail(BSGS_ERR_ARG, "null");
    if (strategy > BSGS_STARTUP_ALLGATHER) return fail(BSGS_ERR_ARG, "strategy %u: 0 broadcast, 1 local build, 2 all-gather", strategy);
    for (int i = 0; i < n; i++) if (!devs[i]) return fail(BSGS_ERR_ARG, "null device %d", i);
    std::vector<bsgs_startup_report> R(n);
    for (auto &r : R) { memset(&r, 0, sizeof r); r.strategy = strategy; }
    const auto t_all = std::chrono::steady_clock::now();
    const uint64_t M = htsz <= 31 ? 1ull << htsz : htsz, line_bytes = layout == BSGS_TABLE_LINES128_LIST ? 128 : 64;
    auto finish = [&](int rc) {
        for (int i = 0; i < n; i++) { R[i].total_s = since(t_all); if (rep) rep[i] = R[i]; }
        return rc;
    };
    if (strategy == BSGS_STARTUP_ALLGATHER && (n == 1 || M % (uint64_t)n)) strategy = n == 1 ? BSGS_STARTUP_LOCAL : BSGS_STARTUP_BROADCAST;     // nothing to gather / slices would not be equal
    for (auto &r : R) r.strategy = strategy;
    int rc = BSGS_OK;
    // the chain scratch (placed by grade against the table just installed) belongs to the start-up and is taken by each engine right after its table: an engine that had
    // a memory group reserved for it hands the unused part back at that point, which the next engine on the same GPU needs (needs the giants: skipped without them)
    auto prepare = [&](int i) {
        if (!devs[i]->g2) return (int)BSGS_OK;
        const auto t0 = std::chrono::steady_clock::now();
        const int r = bsgs_prepare(devs[i]);
        R[i].prepare_s = since(t0);
        return r;
    };
    if (strategy == BSGS_STARTUP_LOCAL) {
        rc = parallel(devs, n, [&](int i) {
            const auto t0 = std::chrono::steady_clock::now();
            int r = bsgs_build_baby_table_ext(devs[i], w, htsz, layout);
            R[i].build_s = since(t0);
            if (r == BSGS_OK) r = prepare(i);
            return r;
        });
        return finish(rc);
    }
    // receive buffers from every engine's own allocator (a table above 40 GiB gets a memory group reserved for the chain scratch first: bsgs_alloc_table_ext_recv)
    std::vector<void *> lines(n, nullptr), ovf(n, nullptr);
    std::vector<uint64_t> cap(n, 0);
    rc = parallel(devs, n, [&](int i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int r = bsgs_alloc_table_ext_recv(devs[i], w, htsz, layout, &lines[i], &ovf[i], &cap[i]);
        R[i].alloc_s = since(t0);
        return r;
    });
    if (rc) return finish(rc);
    bsgs_fabric *F = nullptr;
    rc = bsgs_fabric_open(&F, devs, n, transport);
    if (rc) return finish(rc);
    for (auto &r : R) r.transport = bsgs_fabric_is_rccl(F) ? BSGS_TRANSPORT_RCCL : BSGS_TRANSPORT_PEER;
    uint64_t set_slots = cap[0], over_total = 0;
    if (strategy == BSGS_STARTUP_BROADCAST) {
        uint64_t n_ovf = 0;
        {
            const auto t0 = std::chrono::steady_clock::now();
            rc = bsgs_build_baby_table_ext_device(devs[0], w, htsz, layout, lines[0], ovf[0], cap[0], &n_ovf, &over_total);
            R[0].build_s = since(t0);
        }
        if (rc == BSGS_OK) {
            const auto t0 = std::chrono::steady_clock::now();
            rc = bsgs_fabric_broadcast(F, lines.data(), M * line_bytes, 0);
            if (rc == BSGS_OK) rc = bsgs_fabric_broadcast(F, ovf.data(), n_ovf * 8, 0);
            for (int i = 0; i < n; i++) { R[i].transfer_s = since(t0); R[i].bytes_received = i ? M * line_bytes + n_ovf * 8 : 0; }
        }
        set_slots = n_ovf;
    } else {
        // ALLGATHER: engine i files the buckets [i M / n, (i + 1) M / n) of the table -- in place inside its full line buffer -- and lists their overflow entries
        const uint64_t list_cap = cap[0] / 2;
        std::vector<void *> list(n, nullptr);
        std::vector<uint64_t> n_list(n, 0), over(n, 0);
        rc = parallel(devs, n, [&](int i) {
            HIPCHK(hipSetDevice(devs[i]->id));
            HIPCHK(bsgs_big_malloc(&list[i], std::max<uint64_t>(list_cap, 1) * 8));
            const auto t0 = std::chrono::steady_clock::now();
            const int r = bsgs_build_baby_table_ext_slice(devs[i], w, htsz, layout, lines[i], (uint32_t)i, (uint32_t)n, list[i], list_cap, &n_list[i], &over[i]);
            R[i].build_s = since(t0);
            return r;
        });
        uint64_t total = 0;
        std::vector<uint64_t> first(n, 0);
        for (int i = 0; i < n; i++) { first[i] = total; total += n_list[i]; over_total += over[i]; }
        std::vector<void *> all(n, nullptr);
        if (rc == BSGS_OK && total > list_cap) rc = fail(BSGS_ERR_NOMEM, "overflow lists of the slices: %llu entries, room for %llu", (unsigned long long)total, (unsigned long long)list_cap);
        if (rc == BSGS_OK) {
            const auto t0 = std::chrono::steady_clock::now();
            rc = bsgs_fabric_allgather(F, lines.data(), M / (uint64_t)n * line_bytes);
            // the overflow lists differ in length: every engine's list goes to its place in every engine's concatenation (N broadcasts)
            if (rc == BSGS_OK) rc = parallel(devs, n, [&](int i) {
                HIPCHK(hipSetDevice(devs[i]->id));
                HIPCHK(bsgs_big_malloc(&all[i], std::max<uint64_t>(total, 1) * 8));
                if (n_list[i]) HIPCHK(hipMemcpyAsync((char *)all[i] + first[i] * 8, list[i], n_list[i] * 8, hipMemcpyDeviceToDevice, devs[i]->stream));
                HIPCHK(hipStreamSynchronize(devs[i]->stream));
                return BSGS_OK;
            });
            for (int r = 0; r < n && rc == BSGS_OK; r++) {
                std::vector<void *> at(n);
                for (int i = 0; i < n; i++) at[i] = (char *)all[i] + first[r] * 8;
                rc = bsgs_fabric_broadcast(F, at.data(), n_list[r] * 8, r);
            }
            for (int i = 0; i < n; i++) { R[i].transfer_s = since(t0); R[i].bytes_received = (M - M / (uint64_t)n) * line_bytes + (total - n_list[i]) * 8; }
        }
        if (rc == BSGS_OK) rc = parallel(devs, n, [&](int i) {
            const auto t0 = std::chrono::steady_clock::now();
            const int r = bsgs_build_overflow_set(devs[i], all[i], total, ovf[i], cap[i]);
            R[i].set_s = since(t0);
            return r;
        });
        for (int i = 0; i < n; i++) {
            (void)hipSetDevice(devs[i]->id);
            if (list[i]) (void)bsgs_big_free(list[i]);
            if (all[i]) (void)bsgs_big_free(all[i]);
        }
    }
    bsgs_fabric_close(F);
    if (rc) return finish(rc);
    rc = parallel(devs, n, [&](int i) {
        const auto t0 = std::chrono::steady_clock::now();
        int r = bsgs_install_table_ext_device(devs[i], lines[i], ovf[i], set_slots, over_total, w, htsz, layout);     // validates the table (overflow bound) on every engine
        R[i].install_s = since(t0);
        if (r == BSGS_OK) r = prepare(i);
        return r;
    });
    return finish(rc);
}

// ---- test hook: the fabric on its own -----------------------------------------------------------------------------------------------------------
// `bytes` per engine (a multiple of 8 * n): engine r's buffer is filled with a pattern of its own, then (1) broadcast from root 0 and (2) an in-place all-gather of
// the slices are run over the chosen transport and every engine's buffer is checked ON ITS DEVICE.  mismatches[0] / [1] = words that differ after (1) / (2).  With one
// engine and BSGS_TRANSPORT_RCCL this still creates a (one-rank) communicator and issues both collectives: librccl is loaded, initialised and called next to the
// engine on memory of the engine's allocator -- what a one-GPU lease can exercise of the RCCL route.
static __global__ void fabric_fill_kernel(u64 *buf, u64 n, u64 seed)
{
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) buf[i] = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed;
}
static __global__ void fabric_check_kernel(const u64 *buf, u64 first, u64 n, u64 seed, unsigned long long *bad)
{
    unsigned long long mine = 0;
    for (u64 i = blockIdx.x * (u64)blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) mine += buf[first + i] != (((first + i + 1) * 0x9E3779B97F4A7C15ull) ^ seed);
    if (mine) atomicAdd(bad, mine);
}
extern "C" int bsgs_debug_fabric_selftest(bsgs_dev *const *devs, int n, uint32_t transport, uint64_t bytes, uint64_t mismatches[2], uint32_t *transport_used)
{
    if (!devs || n < 1 || !mismatches || bytes < 8ull * n || bytes % (8ull * n)) return fail(BSGS_ERR_ARG, "bytes must be a positive multiple of 8 * n");
    bsgs_fabric *F = nullptr;
    int rc = bsgs_fabric_open(&F, devs, n, transport);
    if (rc) return rc;
    if (transport_used) *transport_used = bsgs_fabric_is_rccl(F) ? BSGS_TRANSPORT_RCCL : BSGS_TRANSPORT_PEER;
    const u64 words = bytes / 8, slice = words / n;
    std::vector<void *> buf(n, nullptr);
    std::vector<unsigned long long *> bad(n, nullptr);
    auto run = [&]() -> int {
        for (int i = 0; i < n; i++) {
            HIPCHK(hipSetDevice(devs[i]->id));
            HIPCHK(bsgs_lines_malloc(devs[i], &buf[i], bytes));                   // the allocator the bucket lines come from (chunk-mapped above 40 GiB)
            HIPCHK(hipMalloc(&bad[i], 16));
            HIPCHK(hipMemsetAsync(bad[i], 0, 16, devs[i]->stream));
            hipLaunchKernelGGL(fabric_fill_kernel, dim3(1024), dim3(256), 0, devs[i]->stream, (u64 *)buf[i], words, 1000ull + i);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(devs[i]->stream));
        }
        int r = bsgs_fabric_broadcast(F, buf.data(), bytes, 0);
        if (r) return r;
        for (int i = 0; i < n; i++) {                                            // everybody holds engine 0's pattern now
            HIPCHK(hipSetDevice(devs[i]->id));
            hipLaunchKernelGGL(fabric_check_kernel, dim3(1024), dim3(256), 0, devs[i]->stream, (const u64 *)buf[i], (u64)0, words, 1000ull, bad[i]);
            hipLaunchKernelGGL(fabric_fill_kernel, dim3(1024), dim3(256), 0, devs[i]->stream, (u64 *)buf[i], words, 2000ull + i);      // a pattern of its own again
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(devs[i]->stream));
        }
        r = bsgs_fabric_allgather(F, buf.data(), slice * 8);
        if (r) return r;
        mismatches[0] = mismatches[1] = 0;
        for (int i = 0; i < n; i++) {                                            // slice j of every buffer carries engine j's pattern
            HIPCHK(hipSetDevice(devs[i]->id));
            for (int j = 0; j < n; j++)
                hipLaunchKernelGGL(fabric_check_kernel, dim3(1024), dim3(256), 0, devs[i]->stream, (const u64 *)buf[i], (u64)j * slice, slice, 2000ull + j, bad[i] + 1);
            unsigned long long h[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h, bad[i], 16, hipMemcpyDeviceToHost, devs[i]->stream));
            HIPCHK(hipStreamSynchronize(devs[i]->stream));
            mismatches[0] += h[0]; mismatches[1] += h[1];
        }
        return BSGS_OK;
    };
    rc = run();
    const std::string why = rc ? bsgs_last_error() : "";
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(devs[i]->id);
        if (buf[i]) (void)bsgs_big_free(buf[i]);
        if (bad[i]) (void)hipFree(bad[i]);
    }
    bsgs_fabric_close(F);
    return rc ? fail(rc, "%s", why.c_str()) : BSGS_OK;
}
