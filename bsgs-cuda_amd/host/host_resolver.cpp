// host_resolver.cpp -- the tile dispenser (GetJob 1_9_7File.pb:2077-2092) and the hit resolver (checkerThread 3933-4296): htCPU lookups in RAM or in the file,
// the small BSGS that replaces htCPU for extended tables.
#include "host.h"

// centre of tile `index`: P0 + index * PUBADDBIG (what GetJob accumulates one addition at a time, 1_9_7File.pb:2077-2092)
Affine tile_centre(const Shared &S, uint64_t index)
{
    if (!index) return S.walk_p0;
    return hs::point_add(S.walk_p0, hs::point_mul(S.pubadd, hs::fe_from_u64(index)));
}

// GetJob for a batch: hand out `n` consecutive tiles (1_9_7File.pb:2077-2092).  Only the COUNTER advances on the host; the
// centres are derived on the GPU from the tile index (bsgs_enqueue_walk), or by tile_centre() under -hostcentres.
size_t get_jobs(Shared &S, size_t n, std::vector<Tile> &out, int slot)
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
int htcpu_lookup_file(int fd, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max)
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
int htcpu_lookup(const HostBuf &img, uint64_t ht_items, uint64_t key64, uint32_t *pos, int max)
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

bool try_key(const Shared &S, const Scalar &kprime, Scalar &key_out)
{
    const Affine tp = hs::point_mul(hs::G, kprime);
    if (tp.inf || !hs::fe_equal(tp.x, S.findpub.x) || !hs::fe_equal(tp.y, S.findpub.y)) return false;
    const Scalar key = hs::sc_add(kprime, S.start);
    const Affine rp = hs::point_mul(hs::G, key);
    if (rp.inf || !hs::fe_equal(rp.x, S.realpub.x) || !hs::fe_equal(rp.y, S.realpub.y)) return false;
    key_out = key;
    return true;
}

bool resolve_hit(const Shared &S, const PendingHit &hit, Scalar &key_out)
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

void checker_thread(Shared *S)
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
