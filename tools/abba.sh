#!/bin/bash
# tools/abba.sh "<envA>" "<envB>" [bench args]: A B B A A B B A on one box.  Consecutive bench processes on one MI355X alternate
# between two power-management operating points (profiles/r02d_repeat_same_binary_10_runs.log), so a plain A B A B comparison is
# confounded with that alternation; this order gives each variant both positions equally often.
R=${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; shift 2
for v in A B B A A B B A; do
  if [ $v = A ]; then envs=$A; else envs=$B; fi
  env $envs python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --sustain-s ${SUSTAIN:-0} --steps ${STEPS:-20} --warmup 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('$v  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz' % (d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0)))"
done
