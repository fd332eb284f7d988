#!/bin/bash
# round 3, GPU call X: the fuzz parity test, long: 3 seeds x 4000 cases
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
for seed in 1 2 3; do
  ( BSGS_FUZZ_CASES=4000 BSGS_FUZZ_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k fuzz 2>&1 | tail -12 ) > $O/fuzz_seed$seed.log; echo "seed $seed: $(tail -1 $O/fuzz_seed$seed.log)"
done
