#!/bin/bash
# round 3, GPU call R: full GPU suite + smoke on the final code, then the records that changed with the overflow bound (-w 34 bench, config-3 solve), and the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --w 34 --htsz 31 --no-solve --no-pmc > $O/bench_w34.json 2> $O/bench_w34.err
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err
cat $O/pytest_gpu.log; tail -1 $O/smoke.log; tail -2 $O/config3.log
for f in $O/bench_w34.json $O/bench_w30.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], d['chain_scratch']['from_reserved_group'], (d.get('measured_solve') or {}).get('value'))"; done
