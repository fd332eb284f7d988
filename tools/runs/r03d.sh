#!/bin/bash
# round 3, GPU call D: the records of the round: profile pair (plain / kernel trace / PMC passes), bench at -w 26 / -w 34, two ranks on one GPU,
# config-3 solve (80-bit range, -w 34) and config 4 (1000 keys)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r03d > $O/profile_round.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r03d $O/pmc_traffic.json > /dev/null 2>&1
cd $R
python tools/isa_budget.py $O/isa_budget.json > /dev/null 2>&1
python bench.py --w 26 --htsz 25 --no-solve > $O/bench_w26_config2.json 2> $O/bench_w26.err
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
python bench.py --gpus 2 --same-device --w 26 --htsz 25 --no-pmc --no-solve --no-cpu-baseline > $O/bench_two_ranks_same_device_w26.json 2> $O/bench_two_ranks.err
tail -12 $O/profile_round.log
for f in $O/bench_w26_config2.json $O/bench_w34.json $O/bench_two_ranks_same_device_w26.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), d['chain_scratch'], d.get('n_gpus'))
except Exception as e: print('$f', 'FAILED', e)"; done
tail -2 $O/*.err
