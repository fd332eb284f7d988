#!/bin/bash
# round 3, GPU call 4d: QUAD chain as the default: the whole GPU suite, smoke, then ABBA default (13) against the forced pair chain (10) at -w 30 and -w 34
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
STEPS=30 bash tools/abba.sh "BSGS_KERNEL_VARIANT=10" "BSGS_KERNEL_VARIANT=13" > $O/abba_quad_default_w30.log 2>&1; cat $O/abba_quad_default_w30.log
STEPS=20 bash tools/abba.sh "BSGS_KERNEL_VARIANT=10" "BSGS_KERNEL_VARIANT=13" --w 34 --htsz 31 > $O/abba_quad_default_w34.log 2>&1; cat $O/abba_quad_default_w34.log
