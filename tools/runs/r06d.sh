#!/bin/bash
# round 4, GPU call 6d: the rebuilt table builder (bases on the GPU, four waves per SIMD, scatter fused into the generator, coalesced line closing): parity, then times
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -15 ) | tee $O/pytest.log
BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 30 28 2>&1 | tee $O/build_w30.log
BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 34 31 2>&1 | tee $O/build_w34.log
BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 26 25 2>&1 | tee $O/build_w26.log
