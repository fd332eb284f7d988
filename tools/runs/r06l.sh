#!/bin/bash
# round 4, GPU call 6l: chunk-composed lines with ONE address arena (no address is mapped twice): two-engine host x4, the large-table tests x2, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06l; mkdir -p $O; cd $R
export TMPDIR=/tmp
bash tools/runs/r06k.sh 2>&1 | grep "===\|^rc\|KEY\|error\|fault" | tee $O/two_engines_x4.log
for i in 1 2; do timeout 800 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round3.py tests/test_gpu_host.py -m gpu -x -q -k "w34 or 40GiB or config3 or config5 or parked or two_engines" 2>&1 | tail -3; done | tee $O/pytest_large_tables_x2.log
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/pytest_all.log
