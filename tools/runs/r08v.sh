#!/bin/bash
# round 5, GPU call 8v (NOT KEPT: the start-up did not get shorter -- chain scratch 1.24 s against 1.27-1.55 s -- and the step rate was 0.35 % lower, 4 of 4: reverted): the chain scratch of tables above 0.6 of the HBM taken without drawing and grading extra pieces (they always graded alike): the large tables at full size,
# the placement tests, the step rate at 36 * 2^30 points A B A B against the library of r08u (build/exp_r08u), and the start-up of config 3 (key near the start)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08v; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 ) | tee $O/pytest_fullsize_round3.log
OLD="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_r08u/libbsgs_hip.so"; NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 38654705664 --htsz 3221225472 --layout 4 --no-refquirks-leg ) 2>&1 | tee $O/abba_36g_graded_vs_plain_scratch.log
( python tools/config3_run.py 0.02 /tmp/cfg3v "-w auto" ) 2>&1 | tee $O/config3_key_near_the_start.json
