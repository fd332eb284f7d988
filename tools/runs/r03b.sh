#!/bin/bash
# round 3, GPU call B: the whole GPU suite, the FP64 product check, bench
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -60 ) > $O/pytest_gpu.log
MB=$R/bsgs-cuda_amd/build/microbench
$MB dpfcheck > $O/dpfcheck.json 2>&1
OPS="200 202" bash tools/power_ops.sh > $O/power_ops.jsonl 2>&1
( timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err ) ; echo "bench rc $?" >> $O/bench_w30.err
tail -25 $O/pytest_gpu.log; cat $O/dpfcheck.json; cat $O/power_ops.jsonl; cut -c1-400 $O/bench_w30.json; tail -3 $O/bench_w30.err
