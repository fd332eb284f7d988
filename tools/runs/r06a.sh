#!/bin/bash
# round 4, GPU call 6a: the pruned library (three tile kernels instead of thirteen, no two-stream / pooled / streamed modes, tile kernels in their own
# translation units): the whole GPU suite, smoke, and a plain bench line at the headline geometry
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.log
python bench.py --no-pmc --no-solve --no-cpu-baseline --sustain-s 5 2>$O/bench.err | tee $O/bench_w30.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f G  sustained %.2f G  %.2f ms/launch  %s' % (d['value']/1e9, (d['value_sustained'] or 0)/1e9, d['roofline']['avg_launch_ms'], d['roofline']['kernel']))"
