#!/bin/bash
# round 5, GPU call 7f: the GPU suite on the host with job lanes, BASELINE config 4 (1000 keys) with two jobs side by side, the cold solves after the start-up trimming,
# -w 35 quad / pair chain, and BASELINE config 3 for real at Tune's choice (-w 35 on 1.5 * 2^30 lines of 128 bytes)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07f; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) | tee $O/pytest_gpu.log
( python tools/config4_run.py 1000 /tmp/cfg4 ) 2>&1 | tee $O/config4_1000keys_two_lanes.json
python - <<PY 2>&1 | tee $O/cold_solves.log
import json, sys
sys.path.insert(0, "$R")
import bench
m = bench.measured_solve()
print(json.dumps(m))
print("solve", m.get("value"), "cold", (m.get("cold") or {}).get("value"), "cold best", (m.get("cold_best") or {}).get("value"))
for k in ("cold", "cold_best"):
    for ln in (m.get(k) or {}).get("startup_stages", []) + (m.get(k) or {}).get("tune", []): print("  ", k, ln)
PY
run35() {  # name, env
  ( env $2 timeout 900 python bench.py --w 35 --htsz 1610612736 --no-pmc --no-solve --no-cpu-baseline --no-refquirks-leg --sustain-s 5 > $O/bench_w35_$1.json 2> $O/bench_w35_$1.err; echo "w35 $1 rc $?"; tail -2 $O/bench_w35_$1.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w35_$1.json").read().strip().splitlines()[-1]); r=d["roofline"]; p=d["alu"]["power"] or {}
    print("w35 $1: value %.2f G sustained %.2f ms/launch %.3f tpl %d sclk %.0f W %.0f fp hits %d eff keys/s %.3e overfull %d kernel %s" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], p.get("sclk_MHz_mean",0), p.get("socket_W_mean",0), d["false_positive_hits"], d["effective_keys_per_s"], d["config"]["overflow_buckets"], r["kernel"]))
except Exception as e: print("w35 $1 FAILED", e)
PY
  ) 2>&1 | tee -a $O/w35_variants.log
}
run35 quad "X=1"
run35 pair "BSGS_KERNEL_VARIANT=10"
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
