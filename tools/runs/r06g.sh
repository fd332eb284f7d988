#!/bin/bash
# round 4, GPU call 6g (6f again with a G2-cached ceiling whose probes stay random): (1) price the giants' stream: ABBA shipped vs -DBSGS_G2_CACHED_CEILING (every giant read hits one cached KiB; wrong results, right timing);
# (2) the split of FETCH_SIZE into probe / chain / giants with calibration ratios (tools/fetch_breakdown.py)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
{ echo "# A = shipped, B = every giant read served from one cached KiB (-DBSGS_EXPERIMENT -DBSGS_G2_CACHED_CEILING): headline geometry, 30 launches each"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/exp_g2cached/libbsgs_hip.so"; } 2>&1 | tee $O/abba_g2_cached_ceiling.log
python tools/fetch_breakdown.py $O/fetch_breakdown.json 2>&1 | tee $O/fetch_breakdown.log
