#!/bin/bash
# round 3, GPU call 5j: XCD-aware block -> (tile, slice) map for ANY number of blocks per tile (B) against the multiple-of-8 rule (A = -DBSGS_XCD_MAP_MULT8)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) | tee $O/pytest.log
{ echo "# headline geometry (64 blocks per tile): A = multiple-of-8 rule, B = general map"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_oldmap.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" | head -4; } 2>&1 | tee $O/abba_headline.log
{ echo "# -t 256 -b 88 -p 130 -w 29.87 -htsz 28 (22 blocks per tile): A = multiple-of-8 rule (no locality for this geometry), B = general map"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_oldmap.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" -t 256 -b 88 -p 130 --w 29.87 --htsz 28 | head -4; } 2>&1 | tee $O/abba_b88.log
{ echo "# -t 256 -b 138 -p 244 -w 30.25 -htsz 28 (69 blocks per tile)"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_oldmap.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" -t 256 -b 138 -p 244 --w 30.25 --htsz 28 | head -4; } 2>&1 | tee $O/abba_b138.log
