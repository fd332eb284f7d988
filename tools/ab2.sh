#!/bin/bash
# same-box interleaved A/B over bench ARGUMENTS: tools/ab2.sh "<label>=<bench args>;..." [rounds]
R=${GRAFT_REPO_ROOT:-.}
IFS=';' read -ra CFG <<< "$1"
for round in $(seq 1 ${2:-2}); do for c in "${CFG[@]}"; do
  label=${c%%=*}; args=${c#*=}
  python $R/bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$label round $round  %.2f Gsteps/s  %.3f ms/launch  phases %s' % (d['value']/1e9, r['avg_launch_ms'], r.get('probe_phase')))"
done; done
