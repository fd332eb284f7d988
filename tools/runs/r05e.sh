#!/bin/bash
# round 3, GPU call 5e: the tile kernel compiled for THREE waves per SIMD (168 VGPRs; same instruction count) against the shipped four -- is the fourth wave needed?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
{ echo "# A = shipped (4 waves per SIMD, 128 VGPRs), B = -DBSGS_PAIR2_WAVES=3 (3 waves per SIMD, 162 VGPRs, identical instruction count)"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_w3.so"; } 2>&1 | tee $O/abba_three_waves_per_simd.log
{ echo "# the same at -w 34 -htsz 31"; STEPS=20 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_w3.so" --w 34 --htsz 31 | head -4; } 2>&1 | tee $O/abba_three_waves_per_simd_w34.log
