#!/bin/bash
# round 5, GPU call 7v: the host tests with htCPU searched in its file by default (-sf 1 as in the reference), the new CPU-built-files test, config 4 once more
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07v; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests/test_gpu_host.py tests/test_gpu_round5.py tests/test_gpu_round3.py -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_host.log
( python tools/config4_run.py 1000 /tmp/cfg4v ) 2>&1 | tee $O/config4_1000keys_sf1.json
