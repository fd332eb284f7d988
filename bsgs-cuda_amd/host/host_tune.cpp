// host_tune.cpp -- Tune: table sizing advice per GPU and per range.
#include "host.h"

// ---- Tune (1_9_7File.pb:324-431 prints suggested -t -b -p -w -htsz per GPU from free memory and SM count) ----------
// MI355X version: the engine re-batches internally, so -t/-b/-p only set the tile size; -w / -htsz follow from HBM:
// device bytes = 64*2^htsz (bucket lines) + 4*2^htsz + 4*w (htGPU image) + 64*t*b*p (giants) + chain scratch (~8 GiB).

TuneAdvice tune_advice(uint64_t free_bytes)
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

TunePlan tune_plan(uint64_t free_bytes, double range_bits, int n_gpus, uint64_t maxnonce)
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
std::string plan_flags(const TunePlan &p)
{
    char buf[160];
    if (p.htsz_arg > 31 && (p.w & (p.w - 1))) snprintf(buf, sizeof buf, "-w %llu -buckets %u (extended table)", (unsigned long long)p.w, p.htsz_arg);
    else if (p.htsz_arg > 31) snprintf(buf, sizeof buf, "-w %.0f -buckets %u (extended table)", p.w_log2, p.htsz_arg);
    else snprintf(buf, sizeof buf, "-w %.0f -htsz %u%s", p.w_log2, p.htsz_arg, p.ext ? " -ext" : "");
    return buf;
}
void tune(int gpu)
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
