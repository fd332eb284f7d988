#!/bin/bash
# round 3, GPU call 4w: the default bench line of the tree as committed, with its wall-clock time
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04w; mkdir -p $O; cd $R
export TMPDIR=/tmp
t0=$(date +%s.%N)
timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err
t1=$(date +%s.%N)
echo "bench.py wall seconds: $(echo "$t1 - $t0" | bc)" | tee $O/bench_wall.log
python - <<PY
import json
d=json.loads(open("$O/bench_w30.json").read().strip().splitlines()[-1])
print("%.2f G  sustained %.2f G  %.2f ms  frac %.3f  traffic B/step %.1f  solve %s s" % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic_measured_this_run']['bytes_per_step'], d['measured_solve']['value']))
PY
