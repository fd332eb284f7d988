#!/bin/bash
# round 3, GPU call 4k: quad chain with the prefetch AFTER the minus-probe finish (B) against the shipped order (A): fuzz, ABBA, wait counters
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
( BSGS_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "fuzz or whole_tile" 2>&1 | tail -3 ) > $O/pytest.log; cat $O/pytest.log
STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_prev.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so" > $O/abba_prefetch_after_finish.log 2>&1; cat $O/abba_prefetch_after_finish.log
cd /tmp
rm -rf /tmp/rk; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/rk -- python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 --warmup-s 0 --steps 3 --warmup 1 > /dev/null 2>&1 && python $R/tools/rocprof_summary.py pmc /tmp/rk $O/pmc_wait.csv | grep "false, false, true"
