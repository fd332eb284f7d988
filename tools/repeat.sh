#!/bin/bash
# tools/repeat.sh <n> [bench args]: the same bench n times in a row on one box (run-to-run spread; what correlates with it)
R=${GRAFT_REPO_ROOT:-.}
N=$1; shift
for i in $(seq 1 $N); do
  python $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; p=d['alu']['power'] or {}
print('run $i  %.2f G  launch %.2f ms  rnd64 %.0f GB/s  probe-phase %.2f ms  phase1 %.2f ms  sclk %.0f MHz  %.0f W  setup %.1f s  contiguous/plain %.0f/%.0f GiB' % (d['value']/1e9, r['avg_launch_ms'], r['random_read_64B_peak_GBps'], r['probe_phase']['ms_phase3_probes'], r['probe_phase']['ms_phase1_prefix_products'], p.get('sclk_MHz_mean',0), p.get('socket_W_mean',0), d['setup_s'], d['big_buffers_GiB']['physically_contiguous'], d['big_buffers_GiB']['ordinary_pages']))"
done
