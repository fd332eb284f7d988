#!/bin/bash
# round 3, GPU call 4p: whole GPU suite with the narrow batching for small launches; ABBA at the headline launch size (nothing may change there)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 )
cp gpurun_out/route_a_throughput.json $O/ 2>/dev/null; cat $O/route_a_throughput.json
