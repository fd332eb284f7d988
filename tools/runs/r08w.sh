#!/bin/bash
# round 5, GPU call 8w: what one engine spends building under each start-up strategy at -w 34, again with the builder that fills its overflow list by regions
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08w; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python tools/startup_strategy_times.py 34 31 ) 2>&1 | tail -1 | tee $O/startup_strategy_times_w34.json
