#!/usr/bin/env python3
"""Reduce the per-pass summaries written by tools/profile_round.sh into profiles/<tag>_pmc_traffic.json
(the file bench.py reads for roofline.traffic and the ALU-side figures).

  tools/pmc_traffic.py gpurun_out/prof_<tag> profiles/<tag>_pmc_traffic.json [steps_per_launch]

The configuration the passes ran on is read from <dir>/bench_under_rocprofv3_stats.json (the bench line of the same
command) and stored under "config": bench.py reports the per-step figures only for a run of that configuration.
"""
import csv
import json
import os
import sys


def rows(path):
    if not os.path.exists(path):
        return []
    return list(csv.DictReader(open(path)))


import re
PROD = re.compile(r"giant_pair2_kernel<\d, false, (true|false)>")      # production instantiations (PHASE_PROBE = false)


def pick(d, counter, kernel_sub, exclude=None):
    for name in sorted(os.listdir(d)):
        if not name.startswith("rocprofv3_pmc_"):
            continue
        for r in rows(os.path.join(d, name)):
            ok = PROD.search(r["kernel"]) if kernel_sub is PROD else kernel_sub in r["kernel"]
            if r["counter"] == counter and ok and not (exclude and exclude in r["kernel"]):
                return float(r["mean"]), int(r["dispatches"]), r["kernel"]
    return None, 0, None


def main():
    d, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 48 << 25           # the default launch: 48 tiles of 2^25 giant steps
    prod = PROD
    fetch, nf, kname = pick(d, "FETCH_SIZE", "giant_", None)
    fetch, nf, kname = pick(d, "FETCH_SIZE", prod)
    write, _, _ = pick(d, "WRITE_SIZE", prod)
    gups, _, _ = pick(d, "FETCH_SIZE", "mb_gups_kernel")
    res = {
        "source": "tools/profile_round.sh (rocprofv3 --pmc, one counter group per pass, never combined with traces) on `python bench.py "
                  "--no-cpu-baseline`; per-kernel means in profiles/%s_rocprofv3_pmc_*.csv" % os.path.basename(out).split("_pmc_")[0],
        "kernel": kname, "dispatches_averaged": nf, "steps_per_launch": steps,
        "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB; bytes = value * 1024 (guide: HBM section)",
        "fetch_bytes_per_launch": fetch * 1024, "write_bytes_per_launch": write * 1024,
        "fetch_bytes_per_step": fetch * 1024 / steps, "write_bytes_per_step": write * 1024 / steps,
        "algorithmic_bytes_per_step": 64,
    }
    if gups:
        res["calibration"] = {"kernel": "mb_gups_kernel<4> (random 64-byte reads of the same run: 2^28 lines)", "fetch_bytes_measured": gups * 1024,
                              "bytes_expected": float(1 << 34), "ratio": gups * 1024 / float(1 << 34),
                              "note": "random 64-B reads are counted 1.00x; the guide's under-count applies to wide coalesced streams, i.e. to "
                                      "the chain/giant share (< 20 % of the fetch total)"}
    valu, _, _ = pick(d, "SQ_INSTS_VALU", prod)
    if valu:
        res["valu_instructions_per_step"] = valu * 64 / steps
    for key, cname in (("valu_busy_percent", "VALUBusy"), ("salu_busy_percent", "SALUBusy"), ("mem_unit_stalled_percent", "MemUnitStalled"),
                       ("mean_occupancy_waves_per_cu", "MeanOccupancyPerCU")):
        v, _, _ = pick(d, cname, prod)
        if v is not None:
            res[key] = v
    i64, _, _ = pick(d, "SQ_INSTS_VALU_INT64", prod)
    if i64:
        res["valu_int64_instructions_per_step"] = i64 * 64 / steps
    try:                                                # which configuration was profiled (bench.py compares it with its own run)
        line = [x for x in open(os.path.join(d, "bench_under_rocprofv3_stats.json")).read().splitlines() if x.startswith("{")][-1]
        bj = json.loads(line)
        wl = bj["config"]["workload"].split()
        cfg = {"t": int(wl[1]), "b": int(wl[3]), "p": int(wl[5]), "w": float(wl[7]), "htsz": int(wl[9].rstrip(":")),
               "layout": bj["config"]["table_layout"], "variant": os.environ.get("BSGS_KERNEL_VARIANT", "13")}
        res["config"] = cfg
        res["steps_per_launch"] = steps = int(bj["roofline"]["algorithmic_bytes_per_launch"] / 64)
        for k in ("fetch", "write"):
            res["%s_bytes_per_step" % k] = res["%s_bytes_per_launch" % k] / steps
        if valu:
            res["valu_instructions_per_step"] = valu * 64 / steps
        if i64:
            res["valu_int64_instructions_per_step"] = i64 * 64 / steps
    except Exception as e:
        res["config_error"] = repr(e)
    # launch times: of the plain process, of the process under the kernel trace, and of every PMC pass (a counter pass runs 3-5 % slower)
    def launch_ms(name):
        try:
            line = [x for x in open(os.path.join(d, name)).read().splitlines() if x.startswith("{")][-1]
            return json.loads(line)["roofline"]["avg_launch_ms"]
        except Exception:
            return None
    res["avg_launch_ms_plain_process"] = launch_ms("bench_plain.json")
    res["avg_launch_ms_under_kernel_trace"] = launch_ms("bench_under_rocprofv3_stats.json")
    res["avg_launch_ms_under_pmc"] = {n[len("bench_under_pmc_"):-5]: launch_ms(n) for n in sorted(os.listdir(d)) if n.startswith("bench_under_pmc_")}
    res["avg_launch_ms"] = res["avg_launch_ms_under_pmc"].get("VALUBusy") or res["avg_launch_ms_under_pmc"].get("FETCH_SIZE")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])


if __name__ == "__main__":
    main()
