#!/bin/bash
# round 5, GPU call 7a: the tile kernel without its experiment branches and without spills (fe_inv_block4 re-reads its LDS operands behind a barrier: 34 spilled VGPRs / 144 B
# of scratch per lane -> 0 / 0): the GPU suite, then the round-4 library against the new one in alternating processes, then the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) | tee $O/pytest_gpu.log
B=$R/bsgs-cuda_amd/build
( STEPS=20 bash tools/abba.sh "BSGS_LIB_PATH=$B/exp_r04/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" --no-refquirks-leg ) 2>&1 | tee $O/abba_r04_vs_clean_kernel.log
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; echo "bench rc $?"
tail -c 1500 $O/bench_w30.json | head -c 600; echo
python - <<PY
import json
d=json.loads(open("$O/bench_w30.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value %.2f G sustained %.2f G ms/launch %.3f frac %.4f traffic x%s solve %s cold %s" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["frac"], r.get("traffic_over_algorithmic"), d.get("time_to_solve_64bit_range_measured_s"), d.get("cold_time_to_solve_s")))
PY
