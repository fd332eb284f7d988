#!/bin/bash
# round 5, GPU call 9a: the complete bench line (PMC child passes: measured HBM traffic, VALU instructions per step; kernel-trace child) of the 36 * 2^30-point configuration
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r09a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( time python bench.py --no-cpu-baseline --no-solve --no-refquirks-leg --w 38654705664 --htsz 3221225472 --layout 4 > $O/bench_36g_full.json 2> $O/bench_36g_full.err ) 2>&1 | tail -3
python -c "
import json
d=json.loads(open('$O/bench_36g_full.json').read().strip().splitlines()[-1]); r=d['roofline']; t=r.get('traffic_measured_this_run') or {}
print('%.2f G  %.3f ms/launch  frac %.3f  traffic/algorithmic %s  bytes/step %s  VALU/step %s  VALUBusy %s  trace ratio %s' % (d['value']/1e9, r['avg_launch_ms'], r['frac'], r.get('traffic_over_algorithmic'), (r.get('traffic_corrected') or {}).get('bytes_per_step'), t.get('valu_instructions_per_step'), t.get('valu_busy_percent'), (t.get('kernel_trace') or {}).get('ratio_to_parent_ms_per_step')))"
