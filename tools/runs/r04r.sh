#!/bin/bash
# round 3, GPU call 4r: the fuzz with narrow-eligible geometries (150 default cases, then 3000 more with another seed)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04r; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz" 2>&1 | tail -5 ) > $O/pytest_fuzz.log; cat $O/pytest_fuzz.log
( BSGS_FUZZ_CASES=3000 BSGS_FUZZ_SEED=777 timeout 2400 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz" 2>&1 | tail -5 ) > $O/pytest_fuzz_3000.log; cat $O/pytest_fuzz_3000.log
