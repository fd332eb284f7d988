#!/bin/bash
# round 3, GPU call H: the whole GPU suite (verbose tail), then smoke
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "chain scratch in|passed|failed|FAILED|Error|error" | tail -40 ) > $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
cat $O/pytest_gpu.log; tail -2 $O/smoke.log
