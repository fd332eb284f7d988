#!/bin/bash
# round 3, GPU call 5d: ceiling of halving the chain traffic once more (one stored product per EIGHT giants, -DBSGS_OCT_CEILING: unchanged arithmetic, wrong results)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
{ echo "# A = shipped (one stored product per four giants), B = -DBSGS_OCT_CEILING (per eight; arithmetic unchanged, results wrong)"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_oct.so"; } 2>&1 | tee $O/abba_oct_chain_ceiling.log
