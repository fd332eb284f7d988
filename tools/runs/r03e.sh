#!/bin/bash
# round 3, GPU call E: speed CEILINGS of chain-traffic reductions (builds with -DBSGS_QUAD_CEILING / -DBSGS_NOCHAIN_CEILING: wrong results, right timing)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_quadceil.so" > $O/abba_quad_chain_ceiling.log 2>&1
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_nochain.so" > $O/abba_no_chain_ceiling.log 2>&1
cat $O/abba_quad_chain_ceiling.log; echo; cat $O/abba_no_chain_ceiling.log
