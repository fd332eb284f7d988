#!/bin/bash
# round 5, GPU call 8j: where the 10-12 s of start-up of the 36 * 2^30-point table go (stage clocks of the builder and of the placement), host run on a short range
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08j; mkdir -p $O; cd $R
export TMPDIR=/tmp
K=$(python -c "print('%x' % ((1<<70) + 123456789123))")
PB=$(python -c "
import sys; sys.path.insert(0,'bsgs-cuda_amd')
from pybsgs import ecpy
print('%064x%064x' % ecpy.mul((1<<70) + 123456789123))")
( BSGS_BUILD_VERBOSE=1 BSGS_TUNE_VERBOSE=1 ./bsgs-cuda_amd/build/bsgs_mi355x -dir /tmp/r08j -t 256 -b 256 -p 256 -w 38654705664 -buckets 3221225472 -pb $PB -pk $(python -c "print('%x' % (1<<70))") -pke $(python -c "print('%x' % ((1<<70) + (1<<62)))") ) 2>&1 | grep -v "^$" | tail -60 | tee $O/startup_stages_36g.log
