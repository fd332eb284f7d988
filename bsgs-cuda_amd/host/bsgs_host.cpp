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
#include "host.h"

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
    std::set<int> already_won;
    if (!c.recovery_file.empty()) {
        std::ifstream f(c.recovery_file);
        std::string l1, l2, l3, l4;
        auto strip = [](std::string s) { while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back(); return s; };
        if (!std::getline(f, l1) || !std::getline(f, l2) || !std::getline(f, l3) || !std::getline(f, l4)) die("Can`t read recovery file");
        if (strip(l4) != fingerprint(c)) die("Recovery file was made with other settings");
        rec_pos = atoi(strip(l1).c_str()); rec_pub = strip(l2); rec_cnt = strip(l3); recovery = true;
        // list positions win.txt already reports are not searched again (with several lanes a younger job can be reported before the checkpoint names its successor)
        std::ifstream wf(c.dir + "/win.txt", std::ios::binary);
        std::string wl;
        while (std::getline(wf, wl)) if (wl.rfind("KEY[", 0) == 0) already_won.insert(atoi(wl.c_str() + 4));
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
        test_corrupt_engine(devs0);                                   // (test build only)
        if (devs0.size() > 1 && c.verify_replicas) verify_replicas(gpus0, devs0);
        if (c.verify_replicas) verify_tables(S, gpus0, devs0);        // the reference's checkHT / checkHTpackFile / checkGiantArr before it searches (1_9_7File.pb:3717, 3731, 4859, 1941)
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
                while (next_job < pubs.size() && already_won.count((int)next_job + 1)) {      // ... and so is every position win.txt reports already (then the saved counter belongs to a finished job)
                    { std::lock_guard<std::mutex> lo(out_mutex); outs[next_job].done = true; }
                    if (recovery && (int)next_job + 1 == rec_pos) recovery = false;
                    next_job++;
                }
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
            {
                // currentwork.txt must stop naming this job the moment it is over (the timer would let it stand for up to -wt seconds: a restart in that window searched a
                // reported key again and appended a second KEY[n]): it now names the oldest job still in flight, or -- none in flight -- the next list position from its start
                std::lock_guard<std::mutex> lk(lane_mutex);
                lane_listpos[l] = 0;
                int oldest = -1;
                for (size_t q = 0; q < lanes; q++) if (lane_listpos[q] > 0 && (oldest < 0 || lane_listpos[q] < lane_listpos[oldest])) oldest = (int)q;
                if (oldest >= 0) save_checkpoint(*lane_state[oldest]);
                else if (next_job < pubs.size()) {
                    Shared nxt;
                    Affine q;
                    if (hs::parse_pubkey(q, pubs[next_job]) && hs::on_curve(q)) {
                        nxt.cfg = S.cfg; nxt.listpos = (int)next_job + 1; nxt.mainpub_hex = hs::fe_to_hex(q.x) + hs::fe_to_hex(q.y); nxt.glob_key = hs::fe_from_u64(1);
                        save_checkpoint(nxt);
                    }
                }
            }
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
