#!/bin/bash
# round 5, GPU call 7c: the whole GPU suite on the new start-up code (three strategies, fabric, census, any number of buckets, sorted line closing), then -w 34 (what does the
# sorted closing cost the builder?) and the first -w 35 bench with the 48-bit bucket function
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07c; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) | tee $O/pytest_gpu.log
for cfg in "34 31" "35 1610612736"; do
  set -- $cfg
  ( BSGS_BUILD_VERBOSE=1 timeout 900 python bench.py --w $1 --htsz $2 --no-pmc --no-solve --no-cpu-baseline --sustain-s 5 > $O/bench_w$1.json 2> $O/bench_w$1.err; echo "w$1 rc $?"; grep '^\[build\]\|^\[place\]' $O/bench_w$1.err | tail -12; tail -3 $O/bench_w$1.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w$1.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("w$1 value %.2f G sustained %.2f ms/launch %.3f tpl %d build %s setup %.1f fp hits %d eff keys/s %.3e" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], d["table_build"], d["setup_s"], d["false_positive_hits"], d["effective_keys_per_s"]))
    print(d["chain_scratch"], d["config"]["workload"])
except Exception as e: print("w$1 FAILED", e)
PY
  ) 2>&1 | tee $O/w$1.log
done
