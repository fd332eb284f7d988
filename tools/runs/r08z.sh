#!/bin/bash
# round 5, GPU call 8z: smoke() and the whole GPU suite on the FINAL HEAD (the builder changed once more after r08u), then the driver's own command
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08z; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) | tee $O/pytest_gpu.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
