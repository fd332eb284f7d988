#!/bin/bash
# round 3, GPU call 4h: records of the final code after the grader fix: GPU suite, profile pair + PMC passes, ISA budget, bench lines
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04h; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
bash tools/profile_round.sh r04h > $O/profile_round.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r04h $O/pmc_traffic.json > /dev/null 2>&1
cd $R
python tools/isa_budget.py $O/isa_budget.json > /dev/null 2>&1
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err
python bench.py --w 26 --htsz 25 --no-solve --no-pmc > $O/bench_w26_config2.json 2> $O/bench_w26.err
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
tail -4 $O/profile_round.log
for f in $O/bench_w30.json $O/bench_w26_config2.json $O/bench_w34.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); c=d['chain_scratch']; print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], 'graded %d grades %.1f..%.1f' % (c['graded'], c['worst_kept_grade_G_per_s'], c['best_grade_G_per_s']), (d.get('measured_solve') or {}).get('value'))"; done
python -c "
import json; d=json.load(open('$O/pmc_traffic.json')); print({k:d[k] for k in ('fetch_bytes_per_step','write_bytes_per_step','valu_instructions_per_step','valu_busy_percent','avg_launch_ms_plain_process','avg_launch_ms_under_kernel_trace')})"
