#!/bin/bash
# usage: tools/sweep.sh "<variants>" "<tiles-per-launch list>" [extra bench args]   -> one line per combination
R=${GRAFT_REPO_ROOT:-.}
for v in $1; do for n in $2; do
  BSGS_KERNEL_VARIANT=$v python $R/bench.py --no-cpu-baseline --steps 48 --warmup 8 --tiles-per-launch $n $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('var=$v tpl=$n  %.2f Gsteps/s  launch %.3f ms  frac_rand %.3f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac_of_random_read_peak']))"
done; done
