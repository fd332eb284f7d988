#!/bin/bash
# round 5, GPU call 8x: the bench line of the 36 * 2^30-point table (Tune's choice for large ranges) plain and under rocprofv3 --kernel-trace --stats: the tile kernel <4, ., .>
# and the builder's kernels by their own durations
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rpx && mkdir -p /tmp/rpx
ARGS="--no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 10 --steps 20 --warmup 3 --w 38654705664 --htsz 3221225472 --layout 4"
python $R/bench.py $ARGS 2>/tmp/rpx/plain.err | tail -1 > $O/bench_36g_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpx/stats -- python $R/bench.py $ARGS 2>/tmp/rpx/stats.err | tail -1 > $O/bench_36g_under_rocprofv3_stats.json
python $R/tools/rocprof_summary.py stats /tmp/rpx/stats $O/rocprofv3_kernel_stats_36g.csv > /dev/null
head -8 $O/rocprofv3_kernel_stats_36g.csv | cut -c1-220
for f in $O/bench_36g_plain.json $O/bench_36g_under_rocprofv3_stats.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); tb=d.get('table_build') or {}
print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), '%.3f ms/launch' % d['roofline']['avg_launch_ms'], 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9), 'frac %.3f' % d['roofline']['frac'], 'build %.2f s = %.2f G points/s' % (tb.get('seconds',0), tb.get('points_per_s',0)/1e9))"; done
