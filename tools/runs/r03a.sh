#!/bin/bash
# round 3, GPU call A: new tests, multiplier / half-line microbenchmarks with power, first full bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q 2>&1 | tail -40 ) > $O/pytest_round3.log
MB=$R/bsgs-cuda_amd/build/microbench
$MB dpfcheck > $O/dpfcheck.json 2>&1
OPS="200 201 202 0 4 100 105" bash tools/power_ops.sh > $O/power_ops.jsonl 2>&1
$MB 16384 2>&1 | grep -E '"coop"|device' > $O/gups_coop.jsonl
( timeout 900 python bench.py > $O/bench_w30.json 2> $O/bench_w30.err ) ; echo "bench rc $?" >> $O/bench_w30.err
tail -5 $O/pytest_round3.log; cat $O/dpfcheck.json; cat $O/power_ops.jsonl; grep 16384 $O/gups_coop.jsonl | cut -c1-200; cut -c1-1500 $O/bench_w30.json; tail -3 $O/bench_w30.err
