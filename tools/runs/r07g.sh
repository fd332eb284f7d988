#!/bin/bash
# round 5, GPU call 7g: the 128-byte-line kernel with its quad-chain temporaries in registers (10 KiB of LDS per wave: twelve waves per CU) in four- and two-wave blocks,
# parity of that kernel (the suites that run 128-byte lines), BASELINE config 4 with the files written behind the search and 10 / 14 batches per job, the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07g; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "not bench" 2>&1 | tail -6 ) | tee $O/pytest_lines128.log
run35() {  # name, env
  ( env $2 timeout 900 python bench.py --w 35 --htsz 1610612736 --no-pmc --no-solve --no-cpu-baseline --no-refquirks-leg --sustain-s 5 > $O/bench_w35_$1.json 2> $O/bench_w35_$1.err; echo "w35 $1 rc $?"; tail -2 $O/bench_w35_$1.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w35_$1.json").read().strip().splitlines()[-1]); r=d["roofline"]; p=d["alu"]["power"] or {}
    print("w35 $1: value %.2f G sustained %.2f ms/launch %.3f tpl %d sclk %.0f W %.0f fp hits %d eff keys/s %.3e kernel %s normalised %s nJ %s" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], p.get("sclk_MHz_mean",0), p.get("socket_W_mean",0), d["false_positive_hits"], d["effective_keys_per_s"], r["kernel"], d.get("value_clock_normalised"), d.get("nJ_per_giant_step")))
except Exception as e: print("w35 $1 FAILED", e)
PY
  ) 2>&1 | tee -a $O/w35_variants.log
}
run35 treg_block256 "X=1"
run35 treg_block128 "BSGS_LINES128_BLOCK=128"
( python tools/config4_run.py 1000 /tmp/cfg4a ) 2>&1 | tee $O/config4_1000keys_10_batches.json
( BSGS_SHORT_JOB_BATCHES=14 python tools/config4_run.py 1000 /tmp/cfg4b ) 2>&1 | tee $O/config4_1000keys_14_batches.json
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench_w30.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value %.2f G sustained %.2f G ms/launch %.3f frac %.4f traffic x%s solve %s cold %s cold best %s per GHz %.3e normalised %.3e nJ %.2f" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), d.get("time_to_solve_64bit_range_measured_s"), d.get("cold_time_to_solve_s"), d.get("cold_time_to_solve_best_s"), d.get("value_per_GHz") or 0, d.get("value_clock_normalised") or 0, d.get("nJ_per_giant_step") or 0))
PY
