#!/bin/bash
# round 5, GPU call 7e: the unsaturated-limb multipliers on the record (exactness, sustained rate, socket power), the cold solves (config-2 flags and Tune's own choice),
# -w 35 with the pair chain (two probes in flight per wave) and with 1.625 * 2^30 buckets
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07e; mkdir -p $O; cd $R
export TMPDIR=/tmp
MB=$R/bsgs-cuda_amd/build/microbench
( $MB unsatcheck; OPS="200 203 204 205 201" bash tools/power_ops.sh ) 2>&1 | tee $O/modmul_variants.log
python - <<PY 2>&1 | tee $O/cold_solves.log
import json, sys
sys.path.insert(0, "$R")
import bench
m = bench.measured_solve()
print(json.dumps(m))
print("solve", m.get("value"), "cold", (m.get("cold") or {}).get("value"), "cold best", (m.get("cold_best") or {}).get("value"))
for ln in (m.get("cold_best") or {}).get("startup_stages", []) + (m.get("cold_best") or {}).get("tune", []): print("  ", ln)
PY
wait_free() { python - <<PY
import time, torch
for k in range(120):
    fr = torch.cuda.mem_get_info(0)[0]
    if fr > 262 * 2**30: break
    time.sleep(1)
print("free %.1f GiB after %d s" % (fr / 2**30, k))
PY
}
run35() {  # name, env, htsz
  wait_free
  ( env $2 timeout 900 python bench.py --w 35 --htsz $3 --no-pmc --no-solve --no-cpu-baseline --no-refquirks-leg --sustain-s 5 > $O/bench_w35_$1.json 2> $O/bench_w35_$1.err; echo "w35 $1 rc $?"; tail -2 $O/bench_w35_$1.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w35_$1.json").read().strip().splitlines()[-1]); r=d["roofline"]; p=d["alu"]["power"] or {}
    print("w35 $1: value %.2f G sustained %.2f ms/launch %.3f tpl %d sclk %.0f W %.0f fp hits %d eff keys/s %.3e overfull %d kernel %s" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], p.get("sclk_MHz_mean",0), p.get("socket_W_mean",0), d["false_positive_hits"], d["effective_keys_per_s"], d["config"]["overflow_buckets"], r["kernel"]))
except Exception as e: print("w35 $1 FAILED", e)
PY
  ) 2>&1 | tee -a $O/w35_variants.log
}
run35 quad_1p5 "X=1" 1610612736
run35 pair_1p5 "BSGS_KERNEL_VARIANT=10" 1610612736
run35 quad_1p625 "X=1" 1744830464
