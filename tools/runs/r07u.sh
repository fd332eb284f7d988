#!/bin/bash
# round 5, GPU call 7u: the whole GPU suite and smoke() on the final host (table limits, -cpugen, pre-faulted host images, deferred writers) and library (fabric: groups always
# ended, all-gather in 64-bit words)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07u; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) | tee $O/smoke.log
