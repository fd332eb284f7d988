#!/bin/bash
# round 3, GPU call 4m: records of the shipped build at the other operating points: -w 26 (config 2), -w 34, a 60-s sustained region, two ranks on one GPU, config 3 / 4
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
export TMPDIR=/tmp
python bench.py --w 26 --htsz 25 --no-solve --no-pmc > $O/bench_w26_config2.json 2> $O/bench_w26.err
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
python bench.py --sustain-s 60 --no-solve --no-pmc --no-cpu-baseline > $O/bench_w30_60s_sustained.json 2> $O/bench_60s.err
python bench.py --gpus 2 --same-device --w 26 --htsz 25 --no-pmc --no-solve --no-cpu-baseline > $O/bench_two_ranks_same_device_w26.json 2> $O/bench_two_ranks.err
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
( timeout 900 python tools/config4_run.py > $O/config4.log 2>&1 )
for f in $O/bench_w26_config2.json $O/bench_w34.json $O/bench_w30_60s_sustained.json $O/bench_two_ranks_same_device_w26.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['chain_scratch']['pieces'])
except Exception as e: print('$f', 'FAILED', e)"; done
tail -1 $O/config3.log | cut -c1-400; tail -1 $O/config4.log | cut -c1-300
