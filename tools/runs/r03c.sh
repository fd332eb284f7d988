#!/bin/bash
# round 3, GPU call C: half-line speed experiment (variant 12: first 32 bytes of every line only; NOT exact) against the default
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
export TMPDIR=/tmp
STEPS=30 bash tools/abba.sh "BSGS_KERNEL_VARIANT=10" "BSGS_KERNEL_VARIANT=12" > $O/abba_halfline_experiment.log 2>&1
cat $O/abba_halfline_experiment.log
