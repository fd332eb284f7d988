#!/bin/bash
# round 3, GPU call N: bench line with the children run after the parent released its memory
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err ); echo "bench rc $?"
python -c "
import json
d=json.loads(open('$O/bench_w30.json').read().strip().splitlines()[-1])
print('%.2f G  sustained %.2f G  %.3f ms/launch' % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms']))
m=d['roofline']['traffic_measured_this_run']; print(m.get('kernel_trace')); print({k:v.get('child_avg_launch_ms_under_pmc') for k,v in m['passes'].items()}); print(m.get('bytes_per_step'), m.get('valu_busy_percent')); print(d['measured_solve']['value'], d['cpu_baseline']['value'])"
tail -3 $O/bench_w30.err
