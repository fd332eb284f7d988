#!/bin/bash
# round 3, GPU call M: the new round-3 tests, then the bench line with the in-run kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_round3.log
cat $O/pytest_round3.log
( timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err ); echo "bench rc $?"
python -c "
import json
d=json.loads(open('$O/bench_w30.json').read().strip().splitlines()[-1])
print('%.2f G  sustained %.2f G  %.3f ms/launch' % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms']))
m=d['roofline']['traffic_measured_this_run']; print(m.get('kernel_trace')); print(m.get('bytes_per_step'), m.get('valu_busy_percent')); print(d['measured_solve']['value'])"
