#!/bin/bash
# round 3, GPU call 4q: the default batch length at full-size launches: 1024 giants per inversion (A) against 2048 (B) and 512 (C)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04q; mkdir -p $O; cd $R
export TMPDIR=/tmp
{ echo "# A = default (16384 threads x 1024 giants), B = BSGS_BATCH_MULT=8 (8192 x 2048)"; STEPS=30 bash tools/abba.sh "A=1" "BSGS_BATCH_MULT=8"; } 2>&1 | tee $O/abba_batch_2048.log
{ echo "# A = default (16384 threads x 1024 giants), B = BSGS_BATCH_MULT=2 (32768 x 512)"; STEPS=30 bash tools/abba.sh "A=1" "BSGS_BATCH_MULT=2"; } 2>&1 | tee $O/abba_batch_512.log
