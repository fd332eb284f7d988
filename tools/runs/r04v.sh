#!/bin/bash
# round 3, GPU call 4v: final verification of the tree as committed: GPU suite, smoke, default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04v; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
/usr/bin/time -v timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; grep -E "Elapsed" $O/bench_w30.err; cut -c1-300 $O/bench_w30.json
