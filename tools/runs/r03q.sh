#!/bin/bash
# round 3, GPU call Q: what the remaining overflow-set searches cost at -w 34 (ceiling: never search)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
STEPS=20 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_noovf.so" --w 34 --htsz 31 > $O/abba_w34_no_overflow_search_ceiling.log 2>&1
cat $O/abba_w34_no_overflow_search_ceiling.log
