#!/bin/bash
# round 5, GPU call 9b: BASELINE config 4 once more on the final HEAD (r08o ran it on a box that was four per cent slow throughout: 63.2 s)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r09b; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python tools/config4_run.py 1000 /tmp/cfg4b ) 2>&1 | tail -1 | tee $O/config4_1000keys.json
