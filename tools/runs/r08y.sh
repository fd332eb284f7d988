#!/bin/bash
# round 5, GPU call 8y: the whole-table build lends the overflow set's buffer to the sort of the overflow list (30 GiB less to ask the driver for at 36 * 2^30 points):
# every lines + overflow-set test, the large tables at full size, the bench's build seconds at 36 * 2^30 points and the start-up of config 3 (key near the start)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08y; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "(fingerprint or overflow or direct_line or planted or census or three_startup or startup or allgather or any_number_of_buckets or false_positives or fuzz or extended or borrowed) and not extended_table_w3" 2>&1 | grep -E "passed|failed|rror" | tail -4 ) | tee $O/pytest_overflow_sets.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "extended_table_w35 or extended_table_w34" 2>&1 | grep -E "^census|passed|failed|rror" | tail -8 ) | tee $O/pytest_large_tables.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 5 --warmup 2 --w 38654705664 --htsz 3221225472 --layout 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); tb=d['table_build']
print('36g: build %.2f s = %.2f G points/s, allocation + placement %.2f s, %.2f G steps/s' % (tb['seconds'], tb['points_per_s']/1e9, tb['allocation_and_placement_seconds'], d['value']/1e9))"; done 2>&1 | tee $O/bench_36g_build_seconds.log
( python tools/config3_run.py 0.02 /tmp/cfg3y "-w auto" ) 2>&1 | tee $O/config3_key_near_the_start.json
