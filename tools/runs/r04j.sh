#!/bin/bash
# round 3, GPU call 4j: long fuzz on the final build: 5 seeds x 4000 cases with the default (quad chain), 2 seeds with the pair chain forced
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
for seed in 11 12 13 14 15; do
  ( BSGS_FUZZ_CASES=4000 BSGS_FUZZ_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k fuzz 2>&1 | tail -3 ) > $O/fuzz_seed$seed.log; echo "default, seed $seed: $(tail -1 $O/fuzz_seed$seed.log)"
done
for seed in 21 22; do
  ( BSGS_KERNEL_VARIANT=10 BSGS_FUZZ_CASES=4000 BSGS_FUZZ_SEED=$seed timeout 1500 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k fuzz 2>&1 | tail -3 ) > $O/fuzz_v10_seed$seed.log; echo "pair chain forced, seed $seed: $(tail -1 $O/fuzz_v10_seed$seed.log)"
done
