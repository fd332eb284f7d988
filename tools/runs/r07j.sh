#!/bin/bash
# round 5, GPU call 7j: twin engines that share the owner's table (the host's lanes), BASELINE config 4 with them (two and three lanes), wall time of the default bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07j; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_host.py -m gpu -x -q 2>&1 | tail -40 ) | tee $O/pytest_round5_host.log
( python tools/config4_run.py 1000 /tmp/cfg4a ) 2>&1 | tee $O/config4_1000keys_two_lanes_shared_table.json
( python tools/config4_run.py 1000 /tmp/cfg4b "-lanes 3" ) 2>&1 | tee $O/config4_1000keys_three_lanes_shared_table.json
( BSGS_SHORT_JOB_BATCHES=18 python tools/config4_run.py 1000 /tmp/cfg4c "-lanes 3" ) 2>&1 | tee $O/config4_1000keys_three_lanes_18_batches.json
S=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; E=$(date +%s.%N); echo "default bench wall: $(echo "$E - $S" | bc) s, rc $?" | tee $O/bench_default_wall.log
tail -c 600 $O/bench_default.json
