#!/bin/bash
# round 3, GPU call 5b: the host's table-free resolver with 2^24 stored multiples at -w 34 (was 2^22): host tests, config 3
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_host.py -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest_host.log; cat $O/pytest_host.log
( timeout 1200 python tools/config3_run.py 0.5 > $O/config3.log 2>&1 )
tail -1 $O/config3.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('found','job_time_s','process_wall_s_incl_table_build','giant_steps_per_s','checker')})"
grep -i "resolver table" $O/config3.log | head -2
