#!/bin/bash
# round 3, GPU call 4u: 20000 more fuzz cases on the shipped binary (another seed; includes the long-batch geometries whose small launches run on narrow batchings)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04u; mkdir -p $O; cd $R
export TMPDIR=/tmp
( BSGS_FUZZ_CASES=20000 BSGS_FUZZ_SEED=4242 timeout 2700 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz" 2>&1 | tail -5 ) > $O/pytest_fuzz_20000.log; cat $O/pytest_fuzz_20000.log
