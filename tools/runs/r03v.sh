#!/bin/bash
# round 3, GPU call V: 384 tiles per launch (96 GiB of chain scratch in 24 pieces) against the automatic 192
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
export TMPDIR=/tmp
for v in A B B A A B B A; do
  if [ $v = A ]; then tpl=192; st=20; else tpl=384; st=10; fi
  python bench.py --no-cpu-baseline --no-pmc --no-solve --sustain-s 10 --steps $st --warmup 3 --tiles-per-launch $tpl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('$v tpl $tpl  %.2f Gsteps/s  sustained %.2f G  %.3f ms/launch  sclk %.0f MHz  pieces %d' % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['chain_scratch']['pieces']))"
done | tee $O/abba_tiles_per_launch_384.log
