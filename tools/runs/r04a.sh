#!/bin/bash
# round 3, GPU call 4a: one-instruction probe addresses (line index shifted before the crossbar): parity subset, then ABBA at -w 30 against the previous library
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log; cat $O/pytest.log
B=$R/bsgs-cuda_amd/build
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_prev.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" > $O/abba_probe_address_one_instruction.log 2>&1
cat $O/abba_probe_address_one_instruction.log
