#!/bin/bash
# round 5, GPU call 8g: more baby points on the same 3 * 2^30 lines of 64 bytes (192 GiB) now that over-full lines are cheap (overflow fingerprint):
# w = 34 * 2^30 (load 11.33) and 36 * 2^30 (load 12; the overflow set doubles to 32 GiB) against 32 * 2^30 = 2^35 (load 10.67).  Key rate = step rate * 2 w.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08g; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 8 --steps 20 --warmup 3 --w $2 --htsz 3221225472 --layout 4 2>$O/$1.err | tail -1 > $O/$1.json
python -c "
import json
try:
    d=json.loads(open('$O/$1.json').read()); p=d['alu']['power'] or {}
    w=$2 if $2>36 else 2**$2
    print('$1  w %.0f  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz  over-full %s  keys/s %.3e  build %.2f s' % (w, d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['config']['overflow_buckets'], d['value']*2*w, (d.get('table_build') or {}).get('seconds',0)))
except Exception as e:
    print('$1 failed:', e); print(open('$O/$1.err').read()[-1500:])
"; }
( run w32g 35
  run w34g 36507222016
  run w36g 38654705664
  run w32g_again 35 ) 2>&1 | tee $O/more_points_same_lines.log
