#!/bin/bash
# round 3, GPU call 4z: the profile set of the SHIPPED binary (inversion per block): plain / kernel-trace pair + PMC passes, bench lines at -w 26 / -w 34, config 3
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04z; mkdir -p $O; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r04z > $O/profile_round.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r04z $O/pmc_traffic.json > /dev/null 2>&1
cd $R
python bench.py --w 26 --htsz 25 --no-solve --no-pmc > $O/bench_w26_config2.json 2> $O/bench_w26.err
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
tail -4 $O/profile_round.log
for f in $O/bench_w26_config2.json $O/bench_w34.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['chain_scratch']; print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'])"; done
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:d[k] for k in ('fetch_bytes_per_step','write_bytes_per_step','valu_instructions_per_step','valu_busy_percent','avg_launch_ms_plain_process','avg_launch_ms_under_kernel_trace')})"
tail -1 $O/config3.log | cut -c1-420
