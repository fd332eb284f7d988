// host_engines.cpp -- the per-GPU engines of a run: open, load (giants + table; replicas for several engines), verify, and the search thread (cuda() 1_9_7File.pb:2095-2553).
#include "host.h"

// ---- per-GPU driver thread: cuda() 1_9_7File.pb:2095-2553 ---------------------------------------------------------
// devices are opened and loaded once (1_9_7File.pb:2181-2357) and serve every public key of the run.  Only the first device
// takes the giants and the table from the host (or builds the extended table); the others receive replicas device-to-device
// (bsgs_broadcast_tables) instead of the reference's per-GPU upload over PCIe (1_9_7File.pb:2337, 2350).
bsgs_dev *open_dev(int gpu)
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
uint32_t ext_layout(const Config &c, uint64_t free_bytes)
{
    const uint64_t buckets = c.htsz_arg > 31 ? c.htsz_arg : 1ull << c.htsz_arg;
    const double load = (double)c.w / (double)buckets;
    const bool fits128 = 128ull * buckets + (24ull << 30) < free_bytes;
    return load > 12.5 && fits128 ? BSGS_TABLE_LINES128_LIST : BSGS_TABLE_LINES64_LIST;
}
uint32_t transport_code(const Config &c) { return c.transport == "rccl" ? BSGS_TRANSPORT_RCCL : c.transport == "peer" ? BSGS_TRANSPORT_PEER : BSGS_TRANSPORT_AUTO; }
static const char *transport_name(uint32_t t) { return t == BSGS_TRANSPORT_RCCL ? "RCCL over xGMI" : t == BSGS_TRANSPORT_PEER ? "peer copies" : "none"; }
void print_placement(int gpu, size_t gi, bsgs_dev *dev)
{
    uint32_t info[5] = {0, 0, 0, 0, 0}; float grade[2] = {0.f, 0.f};
    CK(bsgs_chain_placement(dev, info, grade));
    printf("GPU #%d engine %zu: chain scratch in %u piece(s) of %u tiles, %u graded, reserved group: %s\n", gpu, gi, info[0], info[1], info[2], info[4] ? "yes" : "no");
}
// fn(gi) for every engine, one host thread per GPU: engines on distinct GPUs run concurrently, engines that share a GPU (-d 0,0) one after the other
void per_gpu(const std::vector<int> &gpus, const std::function<void(size_t)> &fn)
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
void load_engines(const Shared &S, const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs, const HostBuf &htgpu, const HostBuf &g2)
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
void verify_replicas(const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs)
{
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

// TEST BUILD ONLY (bsgs_mi355x_test, -DBSGS_TEST_HOOKS; the shipped host has no such hook): BSGS_TEST_CORRUPT_ENGINE=k[:byte_offset[:xor_mask]] flips bits of one byte of
// engine k's table after the engines were loaded -- the run must then stop in verify_replicas (several engines) or verify_tables (any number)
void test_corrupt_engine(const std::vector<bsgs_dev *> &devs)
{
#ifdef BSGS_TEST_HOOKS
    const char *e = getenv("BSGS_TEST_CORRUPT_ENGINE");
    if (!e) return;
    unsigned long long k = 0, off = 4096 + 5, mask = 0x10;
    sscanf(e, "%llu:%llu:%llu", &k, &off, &mask);
    if (k >= devs.size()) return;
    fprintf(stderr, "BSGS_TEST_CORRUPT_ENGINE=%s: TEST HOOK -- byte %llu of engine %llu's table is XOR-ed with %#llx before the verification (this run must stop)\n", e, off, k, mask);
    CK(bsgs_debug_corrupt_table(devs[k], off, (uint32_t)mask));
#else
    (void)devs;
#endif
}

// ---- the host verifies what it built or loaded, like the reference ---------------------------------------------------------------------------------
// The reference looks sampled keys up in every table it has just built and again in every table it loads (checkHT 1_9_7File.pb:3599-3627, called :3717;
// checkHTpackFile :3101-3134, called :3731 and :4859: 1024+ random k, k*G must be found) and compares 1024 random giants with (i + 1) * ADDPUBG
// (checkGiantArr :1524-1559, called :1941).  Here, per engine and before the first tile:
//   census   one streaming pass over the installed table (bsgs_table_census): entries in lines + overflow set - bound copies == w, no malformed line, no unsorted line
//   babies   1024 sampled k in [1, w] (1, 2, w - 1, w and random ones): low 64 bits of x(k*G), computed on the host, must be found THROUGH THE SHIPPED PROBE
//            (bsgs_table_lookup); 256 sampled k in (w, 2w] must not be (but for 32-bit hash collisions: at most 2 tolerated)
//   htCPU    reference-format tables: the same k must be found with position k - 1 in the htCPU image (RAM or file) the resolver will use
//   giants   1024 sampled i (0, 1, t*b*p - 1 and random ones): the device's giant i (bsgs_sample_g2) == (i + 1) * ADDPUBG computed on the host
// One console line per engine; any failure stops the run.  -noverify skips.  Cost: 0.02-0.1 s (profiles/r10*_host_verification*).
namespace {
struct Samples {
    std::vector<uint64_t> k, key64;     // the first n_in are in [1, w], the rest in (w, 2w]
    size_t n_in = 0;
    std::vector<uint64_t> gi;           // giant numbers
    std::vector<Affine> giant;          // (gi + 1) * ADDPUBG
};
uint64_t splitmix(uint64_t &s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <typename F> void parallel_for(size_t n, const F &f)
{
    const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)32, (n + 31) / 32}));
    std::vector<std::thread> th;
    for (unsigned q = 0; q < nth; q++) th.emplace_back([&, q]() { for (size_t i = n * q / nth; i < n * (q + 1) / nth; i++) f(i); });
    for (auto &t : th) t.join();
}
Samples make_samples(const Shared &S)
{
    const Config &c = S.cfg;
    Samples m;
    uint64_t seed = 0xB5650000ull ^ c.w ^ ((uint64_t)c.htsz_arg << 40);
    const size_t NIN = 1024, NOUT = 256, NG = 1024;
    m.k = {1, c.w};
    if (c.w >= 4) { m.k.push_back(2); m.k.push_back(c.w - 1); }
    while (m.k.size() < NIN) m.k.push_back(1 + splitmix(seed) % c.w);
    m.n_in = m.k.size();
    m.k.push_back(c.w + 1);
    while (m.k.size() < m.n_in + NOUT) m.k.push_back(c.w + 1 + splitmix(seed) % c.w);
    m.key64.resize(m.k.size());
    parallel_for(m.k.size(), [&](size_t i) { m.key64[i] = hs::point_mul(hs::G, hs::sc_from_u128((hs::u128)m.k[i])).x.l[0]; });
    m.gi = {0, S.maxnonce - 1};
    if (S.maxnonce > 2) m.gi.push_back(1);
    while (m.gi.size() < NG) m.gi.push_back(splitmix(seed) % S.maxnonce);
    m.giant.resize(m.gi.size());
    parallel_for(m.gi.size(), [&](size_t i) { m.giant[i] = hs::point_mul(S.addpubg, hs::fe_from_u64(m.gi[i] + 1)); });
    return m;
}
}  // namespace

void verify_tables(const Shared &S, const std::vector<int> &gpus, const std::vector<bsgs_dev *> &devs)
{
    const Config &c = S.cfg;
    const auto t0 = std::chrono::steady_clock::now();
    auto secs = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const Samples m = make_samples(S);
    const double t_samples = secs();
    // htCPU: the image the resolver will use must hold position k - 1 for every sampled k (checkHTpackFile 1_9_7File.pb:3101-3134)
    if (!c.ext) {
        const uint64_t items = 1ull << c.htsz;
        std::atomic<size_t> bad{0};
        std::atomic<uint64_t> first_bad{0};
        parallel_for(m.n_in, [&](size_t i) {
            uint32_t pos[64];
            const int np = S.tab->htcpu_fd >= 0 ? htcpu_lookup_file(S.tab->htcpu_fd, items, m.key64[i], pos, 64) : htcpu_lookup(S.tab->htcpu, items, m.key64[i], pos, 64);
            bool ok = false;
            for (int q = 0; q < std::min(np, 64); q++) ok |= (uint64_t)pos[q] + 1 == m.k[i];
            if (!ok && !bad.fetch_add(1)) first_bad = m.k[i];
        });
        if (bad.load()) die("table verification FAILED: htCPU does not hold position k - 1 for " + std::to_string(bad.load()) + " of " + std::to_string(m.n_in) +
                            " sampled k (first: k = " + std::to_string(first_bad.load()) + "): the HT files do not belong to -w " + std::to_string(c.w) + " or are damaged");
        printf("Table verification: htCPU (%s) holds position k - 1 for %zu of %zu sampled k*G\n", S.tab->htcpu_fd >= 0 ? "file" : "RAM", m.n_in, m.n_in);
    }
    std::vector<std::string> lines(devs.size()), errors(devs.size());
    per_gpu(gpus, [&](size_t gi) {
        bsgs_dev *dev = devs[gi];
        const auto e0 = std::chrono::steady_clock::now();
        auto bad = [&](const std::string &why) { errors[gi] = "GPU #" + std::to_string(gpus[gi]) + " engine " + std::to_string(gi) + ": " + why; };
        uint64_t cs[8] = {0};
        if (bsgs_table_census(dev, cs) != BSGS_OK) return bad(std::string("census: ") + bsgs_last_error());
        if (cs[7] != c.w || cs[4] || cs[5]) {
            char buf[320];
            snprintf(buf, sizeof buf, "census: the table holds %llu entries (%llu in lines + %llu in the overflow set - %llu bound copies) where -w is %llu; %llu malformed lines, %llu unsorted lines",
                     (unsigned long long)cs[7], (unsigned long long)cs[0], (unsigned long long)cs[2], (unsigned long long)cs[3], (unsigned long long)c.w, (unsigned long long)cs[4], (unsigned long long)cs[5]);
            return bad(buf);
        }
        std::vector<uint8_t> found(m.k.size(), 0);
        if (bsgs_table_lookup(dev, m.key64.data(), m.key64.size(), found.data()) != BSGS_OK) return bad(std::string("lookup: ") + bsgs_last_error());
        size_t miss = 0, extra = 0; uint64_t first_miss = 0;
        for (size_t i = 0; i < m.k.size(); i++) {
            if (i < m.n_in && !found[i]) { if (!miss++) first_miss = m.k[i]; }
            if (i >= m.n_in && found[i]) extra++;
        }
        if (miss) return bad(std::to_string(miss) + " of " + std::to_string(m.n_in) + " sampled k*G, k <= w, are NOT found by the probe (first: k = " + std::to_string(first_miss) + ")");
        if (extra > 2) return bad(std::to_string(extra) + " of " + std::to_string(m.k.size() - m.n_in) + " sampled k*G with k > w ARE found: this is not the table of -w " + std::to_string(c.w));
        std::vector<uint8_t> xy(m.gi.size() * 64);
        if (bsgs_sample_g2(dev, m.gi.data(), (uint32_t)m.gi.size(), xy.data()) != BSGS_OK) return bad(std::string("giants: ") + bsgs_last_error());
        for (size_t i = 0; i < m.gi.size(); i++) {
            const Affine g = hs::affine_from_le(&xy[i * 64], &xy[i * 64 + 32]);
            if (!hs::fe_equal(g.x, m.giant[i].x) || !hs::fe_equal(g.y, m.giant[i].y))
                return bad("giant " + std::to_string(m.gi[i]) + " is not " + std::to_string(m.gi[i] + 1) + " * GiantSUBpubkey: Est. " + hs::fe_to_hex(m.giant[i].x) + " - got " + hs::fe_to_hex(g.x));
        }
        char buf[400];
        snprintf(buf, sizeof buf, "Table verification: GPU #%d engine %zu: census %llu = -w (%llu over-full lines, %llu set keys), %zu/%zu sampled k*G found, %zu/%zu beyond w, %zu giants = (i+1)*GiantSUBpubkey, %.3fs",
                 gpus[gi], gi, (unsigned long long)cs[7], (unsigned long long)cs[1], (unsigned long long)cs[2], m.n_in, m.n_in, extra, m.k.size() - m.n_in, m.gi.size(),
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - e0).count());
        lines[gi] = buf;
    });
    for (size_t gi = 0; gi < devs.size(); gi++) {
        if (!errors[gi].empty()) die("table verification FAILED: " + errors[gi]);
        printf("%s\n", lines[gi].c_str());
    }
    printf("[startup] %-44s %.3fs (samples on the host %.3fs)\n", "table + giants verification", secs(), t_samples);
}

void gpu_thread(Shared *S, int gpu, int slot, bsgs_dev *dev)
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
