#!/bin/bash
# round 3, GPU call 4s: three more ceilings / settings against the shipped build: the second reads of Gx served from one cached KiB (-DBSGS_G2_DUP_CEILING),
# tiles per G2-sharing chunk 32 and 128 (shipped: 64)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04s; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
for nm in g2dup chunk32 chunk128; do
  { echo "# A = shipped, B = libbsgs_hip_$nm.so"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_$nm.so"; } 2>&1 | tee $O/abba_$nm.log
done
