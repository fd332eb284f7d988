// host_selftest.cpp -- -selftest: the host-side logic that needs no GPU (CPU test tier, tests/test_host_logic.py).
#include "host.h"

// ---- -selftest: the host-side logic that needs no GPU (CPU test tier, tests/test_host_logic.py) ------------------------
// prints "key value" lines: SHA1, the configuration fingerprint, host EC arithmetic, public-key parsing, the dispenser
// sequence and the table-free resolver, each for the inputs given on the command line
int selftest(int argc, char **argv)
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
