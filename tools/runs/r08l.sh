#!/bin/bash
# round 5, GPU call 8l: the 36 * 2^30-point table leaves 62 GiB free, the launch-size rule (chain scratch within a third of the free memory) then takes 96 tiles per launch:
# the same table with 192-tile launches (24 GiB of scratch) and 128, A B A
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08l; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 8 --steps 20 --warmup 3 --w 38654705664 --htsz 3221225472 --layout 4 --tiles-per-launch $2 2>$O/$1.err | tail -1 > $O/$1.json
python -c "
import json
try:
    d=json.loads(open('$O/$1.json').read()); p=d['alu']['power'] or {}
    print('$1  tiles/launch $2  %.2f Gsteps/s  sustained %.2f  %.3f ms/launch  sclk %.0f MHz' % (d['value']/1e9, (d.get('value_sustained') or 0)/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0)))
except Exception as e:
    print('$1 failed:', e); print(open('$O/$1.err').read()[-1500:])
"; }
( run auto_a 0
  run t192_a 192
  run t128 128
  run auto_b 0
  run t192_b 192 ) 2>&1 | tee $O/launch_size_36g.log
