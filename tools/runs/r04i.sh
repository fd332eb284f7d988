#!/bin/bash
# round 3, GPU call 4i: the final build once more: smoke, the driver's bench command, the GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04i; mkdir -p $O; cd $R
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; m=r['traffic_measured_this_run']
print('%.2f G  sustained %.2f G  %.3f ms/launch  frac %.3f  frac_alu %s  traffic %.1f B/step  solve %s s  kernel %s' % (d['value']/1e9, d['value_sustained']/1e9, r['avg_launch_ms'], r['frac'], r['frac_alu'], (m.get('bytes_per_step') or 0), d['measured_solve']['value'], r['kernel']))
print(m.get('kernel_trace'))"
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 )
