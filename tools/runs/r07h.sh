#!/bin/bash
# round 5, GPU call 7h: which test of the 128-byte-line subset failed in r07g (with the details this time), what one engine of an 8-GPU start-up spends building under
# each strategy (-w 34), BASELINE config 4 once more (host images without zero-fill, 14 batches per job), BASELINE config 3 at Tune's choice on the register-temporaries kernel
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07h; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "not bench" 2>&1 | tail -60 ) | tee $O/pytest_lines128.log
( python tools/startup_strategy_times.py 34 31 ) 2>&1 | tee $O/startup_strategy_times_w34.json
( python tools/config4_run.py 1000 /tmp/cfg4c ) 2>&1 | tee $O/config4_1000keys.json
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
