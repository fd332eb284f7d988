#!/bin/bash
# round 5, GPU call 8i: the overflow set as a MULTISET of the overflow list (r08h: the census of the 36 * 2^30-point table came out 4 short -- identical (bucket, hash) pairs folded
# into one key): the full-size tests of the large tables again, then every test that touches an overflow set
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08i; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "extended_table_w35 or extended_table_w34" 2>&1 | grep -v "^\[build\]" | tail -12 ) | tee $O/pytest_large_tables.log
( timeout 1500 python -m pytest tests -m gpu -x -q -k "(fingerprint or overflow or direct_line or planted or census or three_startup or any_number_of_buckets or false_positives or fuzz or extended) and not extended_table_w3" 2>&1 | tail -6 ) | tee $O/pytest_overflow_sets.log
