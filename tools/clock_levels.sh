#!/bin/bash
# tools/clock_levels.sh <runs>: every clock domain rocm-smi reports (sclk, mclk, fclk, socclk) sampled under the tile kernel, run after
# run, next to the launch time of that run: do the run-to-run levels show in a clock other than sclk?
R=${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-4}); do
  python $R/bench.py --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null > /tmp/clk_bench.json &
  BP=$!
  while kill -0 $BP 2>/dev/null; do
    rocm-smi -c --showpower --json 2>/dev/null | python -c "
import json,sys,re
d=json.load(sys.stdin)['card0']
w=[float(v) for k,v in d.items() if 'Power' in k][0]
if w > 900:
    print('    %4.0f W ' % w, ' '.join('%s %s' % (k.split(' ')[0], re.sub(r'[()]', '', v)) for k, v in sorted(d.items()) if 'clock speed' in k.lower()))"
    sleep 0.5
  done
  python -c "
import json
d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1])
print('run $i  %.2f G  launch %.2f ms' % (d['value']/1e9, d['roofline']['avg_launch_ms']))"
done
