#!/bin/bash
# round 3, GPU call 4l: the final build (prefetch after the minus-probe finish): whole GPU suite, smoke, default bench line with in-run PMC
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04l; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/smoke.log; cat $O/smoke.log
timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; cat $O/bench_w30.json
