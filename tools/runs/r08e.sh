#!/bin/bash
# round 5, GPU call 8e: the whole GPU suite and smoke() on HEAD after the 31-bit overflow fingerprint and the host-test fix (r08d stopped at that test)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08e; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $O/smoke.log
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_gpu.log
