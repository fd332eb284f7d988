#!/bin/bash
# round 3, GPU call 4c: QUAD chain (variant 13: one stored product per four giants, one probe in flight): parity (variants test, fuzz under the variant), then ABBA against the pair kernel
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04c; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants" 2>&1 | tail -4 ) > $O/pytest_variants.log; cat $O/pytest_variants.log
( BSGS_KERNEL_VARIANT=13 BSGS_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz" 2>&1 | tail -4 ) > $O/pytest_fuzz_v13.log; cat $O/pytest_fuzz_v13.log
STEPS=30 bash tools/abba.sh "BSGS_KERNEL_VARIANT=10" "BSGS_KERNEL_VARIANT=13" > $O/abba_quad_chain.log 2>&1
cat $O/abba_quad_chain.log
