#!/bin/bash
# round 3, GPU call 4f: the records of the final code (quad chain default, scratch 8 bytes per giant): GPU suite, smoke, profile pair + PMC passes, bench lines, config 3 / 4
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r04f > $O/profile_round.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r04f $O/pmc_traffic.json > /dev/null 2>&1
cd $R
python tools/isa_budget.py $O/isa_budget.json > /dev/null 2>&1
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err
python bench.py --w 26 --htsz 25 --no-solve --no-pmc > $O/bench_w26_config2.json 2> $O/bench_w26.err
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
python bench.py --gpus 2 --same-device --w 26 --htsz 25 --no-pmc --no-solve --no-cpu-baseline > $O/bench_two_ranks_same_device_w26.json 2> $O/bench_two_ranks.err
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
( timeout 900 python tools/config4_run.py > $O/config4.log 2>&1 )
tail -6 $O/profile_round.log
for f in $O/bench_w30.json $O/bench_w26_config2.json $O/bench_w34.json $O/bench_two_ranks_same_device_w26.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['chain_scratch']['pieces'], (d.get('measured_solve') or {}).get('value'))
except Exception as e: print('$f', 'FAILED', e)"; done
tail -1 $O/config3.log | cut -c1-400; tail -1 $O/config4.log | cut -c1-300
