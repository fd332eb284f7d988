#!/bin/bash
# usage: tools/sweep.sh "<variants>" "<tiles-per-launch list>" "<batch mult list>" [extra bench args]
R=${GRAFT_REPO_ROOT:-.}
for m in $3; do for v in $1; do for n in $2; do
  BSGS_BATCH_MULT=$m BSGS_KERNEL_VARIANT=$v python $R/bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 3 --tiles-per-launch $n $4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mult=$m var=$v tpl=$n  %.2f Gsteps/s  launch %.3f ms (%d launches) frac_rand %.3f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['frac_of_random_read_peak']))"
done; done; done
