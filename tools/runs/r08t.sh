#!/bin/bash
# round 5, GPU call 8t: the overflow list filled in one region per block of the generator (a counter each) instead of through ONE counter (r08s: 2.2 s of the 4.9 s scatter at
# 36 * 2^30 points): the tests that build lines + overflow-set tables (every strategy, slices, fuzz), the large tables at full size, the builder's stage clocks, the start-up
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08t; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "(fingerprint or overflow or direct_line or planted or census or three_startup or startup or allgather or any_number_of_buckets or false_positives or fuzz or extended or borrowed) and not extended_table_w3" 2>&1 | tail -6 ) | tee $O/pytest_overflow_sets.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "extended_table_w35 or extended_table_w34" 2>&1 | grep -v "^\[build\]\|chain scratch" | tail -8 ) | tee $O/pytest_large_tables.log
for w in 38654705664 35; do
  echo "== w $w"
  BSGS_BUILD_VERBOSE=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 2 --warmup 1 --warmup-s 0 --w $w --htsz 3221225472 --layout 4 2>&1 | grep -E "^\[build\]|rror" | head -8
done 2>&1 | tee $O/builder_stages.log
( BSGS_BUILD_VERBOSE=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 2 --warmup 1 --warmup-s 0 --w 34 --htsz 31 2>&1 | grep -E "^\[build\]|rror" | head -8 ) 2>&1 | tee -a $O/builder_stages.log
( python tools/config3_run.py 0.02 /tmp/cfg3t "-w auto" ) 2>&1 | tee $O/config3_key_near_the_start.json
