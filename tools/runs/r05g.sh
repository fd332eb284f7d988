#!/bin/bash
# round 3, GPU call 5g: after the narrow_pi refactor: the small-launch tests, the fuzz, smoke
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -x -q -k "small_launches or whole_tile or fuzz or route_a or compat" 2>&1 | tail -4 ) | tee $O/pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
