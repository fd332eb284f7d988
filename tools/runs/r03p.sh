#!/bin/bash
# round 3, GPU call P: overflow bound with SORTED overflow entries (the line keeps the smallest hashes): parity, then ABBA at -w 34 against the previous library
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py tests/test_gpu_host.py -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest.log
cat $O/pytest.log
B=$R/bsgs-cuda_amd/build
STEPS=20 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_prev.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" --w 34 --htsz 31 > $O/abba_w34_overflow_bound_sorted.log 2>&1
cat $O/abba_w34_overflow_bound_sorted.log
