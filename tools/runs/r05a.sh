#!/bin/bash
# round 3, GPU call 5a: 12000 more fuzz cases on the shipped binary (inversion per block; another seed)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( BSGS_FUZZ_CASES=12000 BSGS_FUZZ_SEED=27182 timeout 2400 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz" 2>&1 | tail -4 ) > $O/pytest_fuzz_12000.log; cat $O/pytest_fuzz_12000.log
