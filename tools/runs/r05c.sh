#!/bin/bash
# round 3, GPU call 5c: last verification of the tree as committed: GPU suite, smoke, default bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err
python - <<PY
import json
d=json.loads(open("$O/bench_w30.json").read().strip().splitlines()[-1])
print("%.2f G  sustained %.2f G  %.2f ms  frac %.3f  traffic B/step %.1f  VALU/step %.1f  solve %s s" % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic_measured_this_run']['bytes_per_step'], d['roofline']['traffic_measured_this_run']['valu_instructions_per_step'], d['measured_solve']['value']))
PY
