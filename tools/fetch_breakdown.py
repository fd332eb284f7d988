#!/usr/bin/env python3
"""tools/fetch_breakdown.py OUT.json -- split the tile kernel's FETCH_SIZE into its three streams.

Three counter passes (rocprofv3 --pmc FETCH_SIZE, never combined with a trace) over the same child run of bench.py (settled, last 3 dispatches of the
production kernel), one per library:
    shipped                                  probe lines + stored products (chain) + giants
    build/exp_nochain   (-DBSGS_NOCHAIN_CEILING)     no chain stores, no chain fetches           -> chain  = shipped - this
    build/exp_g2cached  (-DBSGS_G2_CACHED_CEILING)   every giant read served from one cached KiB -> giants = shipped - this
and probe = the rest.  The two experiment libraries return wrong hit lists by construction (bsgs_build_info says so); only their counters are used.
Build them first (round 5: the experiments are written into a COPY of csrc/, the shipped kernel has none):
    tools/experiments/build_experiment.sh nochain "-DBSGS_EXPERIMENT -DBSGS_NOCHAIN_CEILING" ; tools/experiments/build_experiment.sh g2cached "-DBSGS_EXPERIMENT -DBSGS_G2_CACHED_CEILING"."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROD = re.compile(r"giant_pair2_kernel<\d, false, (true|false)>")
B = os.path.join(ROOT, "bsgs-cuda_amd", "build")
LIBS = {"shipped": os.path.join(B, "libbsgs_hip.so"), "no_chain": os.path.join(B, "exp_nochain", "libbsgs_hip.so"), "giants_cached": os.path.join(B, "exp_g2cached", "libbsgs_hip.so")}
CAL = {"mb_gups_kernel<4>": "random_64B_lines", "mb_stream_read_kernel": "coalesced_16B_loads", "mb_stream_read_lds_kernel": "coalesced_16B_lds_dma", "mb_stream_write_nt_kernel": "nt_16B_stores"}


def one_pass(lib, counter, extra):
    tmp = tempfile.mkdtemp(prefix="bsgs_fb_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", "3", "--warmup", "1"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, BSGS_LIB_PATH=lib, TMPDIR="/tmp"), cwd="/tmp")
        files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode or not files:
            raise SystemExit("pass failed (%s, %s): rc %d %s" % (lib, counter, r.returncode, r.stderr[-400:]))
        child = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        with open(files[0]) as f:
            rows = sorted(csv.DictReader(f), key=lambda row: int(row.get("Dispatch_Id", 0) or 0))
        prod = [float(row["Counter_Value"]) for row in rows if PROD.search(row["Kernel_Name"]) and row["Counter_Name"] == counter][-3:]
        cal = {}
        for row in rows:
            for sub, name in CAL.items():
                if sub in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    cal[name] = float(row["Counter_Value"]) * 1024 / float(1 << 34)
        steps = child["roofline"]["algorithmic_bytes_per_launch"] / 64
        return {"bytes_per_step": sum(prod) / len(prod) * 1024 / steps, "dispatches": len(prod), "ms_per_launch_under_pmc": child["roofline"]["avg_launch_ms"],
                "settle_launches": child.get("settle_launches"), "library_build_info": child.get("library_build_info"), "calibration": cal}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    out, extra = sys.argv[1], sys.argv[2:]
    res = {"how": __doc__.split("\n\n")[1], "FETCH_SIZE": {}, "WRITE_SIZE": {}}
    for name, lib in LIBS.items():
        res["FETCH_SIZE"][name] = one_pass(lib, "FETCH_SIZE", extra)
        print(name, res["FETCH_SIZE"][name], flush=True)
    res["WRITE_SIZE"]["shipped"] = one_pass(LIBS["shipped"], "WRITE_SIZE", extra)
    f = {k: v["bytes_per_step"] for k, v in res["FETCH_SIZE"].items()}
    chain, giants = f["shipped"] - f["no_chain"], f["shipped"] - f["giants_cached"]
    probe = f["shipped"] - chain - giants
    res["raw_fetch_B_per_step"] = {"total": f["shipped"], "probe": probe, "chain": chain, "giants": giants}
    res["fractions_of_raw_fetch"] = {"probe": probe / f["shipped"], "chain": chain / f["shipped"], "giants": giants / f["shipped"]}
    cal = res["FETCH_SIZE"]["shipped"]["calibration"]
    cal["nt_16B_stores"] = res["WRITE_SIZE"]["shipped"]["calibration"].get("nt_16B_stores")
    res["calibration_ratios"] = cal
    r = lambda k: cal.get(k) or 1.0      # noqa: E731
    res["corrected_B_per_step"] = {"probe": probe / r("random_64B_lines"), "chain": chain / r("coalesced_16B_lds_dma"), "giants": giants / r("coalesced_16B_loads"),
                                   "write": res["WRITE_SIZE"]["shipped"]["bytes_per_step"] / r("nt_16B_stores")}
    res["corrected_B_per_step"]["total"] = sum(res["corrected_B_per_step"].values())
    res["corrected_non_probe_fetch_B_per_step"] = res["corrected_B_per_step"]["chain"] + res["corrected_B_per_step"]["giants"]
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: res[k] for k in ("raw_fetch_B_per_step", "fractions_of_raw_fetch", "calibration_ratios", "corrected_B_per_step")}))


if __name__ == "__main__":
    main()
