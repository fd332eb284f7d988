#!/bin/bash
# round 3, GPU call 4b: ONE probe in flight per wave (minus probe finished before the plus probe is issued into the same slot: frees 4 KiB of LDS per wave): exact; fuzz parity, then ABBA
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
( BSGS_LIB_PATH=$B/libbsgs_hip_oneslot.so timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz or whole_tile" 2>&1 | tail -4 ) > $O/pytest.log; cat $O/pytest.log
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_oneslot.so" > $O/abba_one_probe_slot.log 2>&1
cat $O/abba_one_probe_slot.log
