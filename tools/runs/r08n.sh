#!/bin/bash
# round 5, GPU call 8n: two fingerprint bits per set-only hash, both tested by the kernels of tables with any number of buckets and by the 128-byte-line kernels: the tests
# that touch lines + overflow-set tables, then A B B A against the one-bit library (build/exp_fp1bit = HEAD before the change) on the 36 * 2^30-point table and at 2^35
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08n; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "(fingerprint or overflow or direct_line or planted or census or three_startup or any_number_of_buckets or false_positives or fuzz or extended) and not extended_table_w3" 2>&1 | tail -6 ) | tee $O/pytest_overflow_sets.log
OLD="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_fp1bit/libbsgs_hip.so"; NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 38654705664 --htsz 3221225472 --layout 4 --no-refquirks-leg ) 2>&1 | tee $O/abba_36g_fp1_vs_fp2.log
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 35 --htsz 3221225472 --layout 4 --no-refquirks-leg ) 2>&1 | tee $O/abba_w35_fp1_vs_fp2.log
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "extended_table_w35 or extended_table_w34" 2>&1 | grep -v "^\[build\]" | tail -8 ) | tee $O/pytest_large_tables.log
