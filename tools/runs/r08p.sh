#!/bin/bash
# round 5, GPU call 8p: -w 34 -htsz 31 (BASELINE config 3 as worded) before and after the two-bit fingerprint (the kernel of 2^htsz-bucket tables tests the first bit only, the
# builder now sets two per hash): r08o found that key in 234.7 s at 37.5 G on a box that also ran config 4 four per cent slow -- A B B A on one box, old = build/exp_fp1bit
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08p; mkdir -p $O; cd $R
export TMPDIR=/tmp
OLD="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_fp1bit/libbsgs_hip.so"; NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 34 --htsz 31 --no-refquirks-leg ) 2>&1 | tee $O/abba_w34_fp1_vs_fp2.log
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --no-refquirks-leg ) 2>&1 | tee $O/abba_w30_fp1_vs_fp2.log
