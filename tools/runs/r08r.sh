#!/bin/bash
# round 5, GPU call 8r: the host tests with the final host binary (one more Tune line at start-up)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08r; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_host.py tests/test_gpu_round5.py tests/test_gpu_round3.py -m gpu -q 2>&1 | tail -8 ) | tee $O/pytest_host.log
( ./bsgs-cuda_amd/build/bsgs_mi355x -h 2>&1 | head -3; mkdir -p /tmp/r08r; ./bsgs-cuda_amd/build/bsgs_mi355x -dir /tmp/r08r -w 20 -htsz 18 -pb 0379be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798 -pk 1 -pke 100000000 2>&1 | grep -i "GPU #0" | head -5 ) | tee $O/tune_lines.log
