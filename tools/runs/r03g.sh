#!/bin/bash
# round 3, GPU call G: full GPU suite after the host changes, then the config-3 (80-bit range, -w 34) and config-4 (1000 keys) records
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 ) > $O/pytest_gpu.log
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
( timeout 900 python tools/config4_run.py > $O/config4.log 2>&1 )
tail -8 $O/pytest_gpu.log; tail -5 $O/config3.log; tail -5 $O/config4.log; ls gpurun_out | tail -5
