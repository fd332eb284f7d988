#!/bin/bash
# round 5, GPU call 8c: the OVERFLOW FINGERPRINT (header of an over-full line = 0xFFFF0000 | bits of the set-only hashes): tests that touch lines + overflow-set tables,
# then A B B A at -w 35 (3 * 2^30 lines of 64 bytes: load 10.67) and at -w 34 (load 8) against the library of the commit before (build/exp_before_fp)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08c; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "fingerprint or overflow or direct_line or planted or census or extended_table or three_startup or any_number_of_buckets or false_positives or fuzz" 2>&1 | tail -15 ) | tee $O/pytest_fingerprint.log
OLD="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_before_fp/libbsgs_hip.so"; NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 35 --htsz 3221225472 --no-refquirks-leg ) 2>&1 | tee $O/abba_w35_fingerprint.log
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 34 --htsz 31 --no-refquirks-leg ) 2>&1 | tee $O/abba_w34_fingerprint.log
