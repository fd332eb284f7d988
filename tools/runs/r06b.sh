#!/bin/bash
# round 4, GPU call 6b: replica verification (checksums, corrupted replica -> red, host -d 0,0), overflow-bound validation, the bench tests that carry the new fields
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -25 ) | tee $O/pytest.log
