#!/bin/bash
# same-box interleaved A/B: tools/ab.sh "<label>=<ENV assignments>;..." [rounds] [bench args]
# e.g. tools/ab.sh "v1=BSGS_KERNEL_VARIANT=1;v6=BSGS_KERNEL_VARIANT=6" 3
R=${GRAFT_REPO_ROOT:-.}
IFS=';' read -ra CFG <<< "$1"
for round in $(seq 1 ${2:-3}); do for c in "${CFG[@]}"; do
  label=${c%%=*}; envs=${c#*=}
  env $envs python $R/bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label round $round  %.2f Gsteps/s  %.3f ms/launch' % (d['value']/1e9, d['roofline']['avg_launch_ms']))"
done; done
