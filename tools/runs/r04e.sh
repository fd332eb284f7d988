#!/bin/bash
# round 3, GPU call 4e: ABBA default quad chain (13) vs forced pair chain (10) at -w 30 once more (another box), longer timed regions
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
export TMPDIR=/tmp
STEPS=40 SUSTAIN=10 bash tools/abba.sh "BSGS_KERNEL_VARIANT=10" "BSGS_KERNEL_VARIANT=13" > $O/abba_quad_default_w30_$(hostname | tail -c 6).log 2>&1; cat $O/abba_quad_default_w30_*.log
