#!/bin/bash
# round 3, GPU call S: a 60-second sustained region (the bench's own sampler records power and clock)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
export TMPDIR=/tmp
python bench.py --sustain-s 60 --no-cpu-baseline --no-pmc --no-solve > $O/bench_60s_sustained.json 2> $O/err.log
python -c "
import json
d=json.loads(open('$O/bench_60s_sustained.json').read().strip().splitlines()[-1]); s=d['sustained']
print('%.2f G burst, %.2f G over %.1f s (%d launches)' % (d['value']/1e9, s['value']/1e9, s['seconds'], s['launches_per_gpu']), s['power'])"
