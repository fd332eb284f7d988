#!/bin/bash
# round 3, GPU call F: which half of the chain traffic costs what (ceilings: no stores / no fetches)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_NOCHAIN_STORE.so" > $O/abba_no_chain_stores_ceiling.log 2>&1
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_NOCHAIN_LOAD.so" > $O/abba_no_chain_fetches_ceiling.log 2>&1
cat $O/abba_no_chain_stores_ceiling.log; echo; cat $O/abba_no_chain_fetches_ceiling.log
