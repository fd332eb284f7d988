#!/bin/bash
# round 5, GPU call 8m: what is left in the overflow-set path at 36 * 2^30 points (load 12): the shipped library against the NO_OVF_CEILING experiment library (never asks
# the set: WRONG results, same timing otherwise), A B B A.  Prices a second fingerprint bit per hash (Bloom, k = 2) before it is written.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08m; mkdir -p $O; cd $R
export TMPDIR=/tmp
NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"; CEIL="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_noovf/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$NEW" "$CEIL" --w 38654705664 --htsz 3221225472 --layout 4 --no-refquirks-leg ) 2>&1 | tee $O/abba_36g_shipped_vs_no_overflow_set.log
