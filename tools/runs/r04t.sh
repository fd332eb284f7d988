#!/bin/bash
# round 3, GPU call 4t: one bench rank under RCCL (BSGS_DIST_FORCE=1): every collective of the N > 1 path next to the engine
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04t; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "one_rank_under_rccl or two_ranks" 2>&1 | tail -25 ) > $O/pytest.log; cat $O/pytest.log
BSGS_DIST_FORCE=1 timeout 600 python bench.py --no-cpu-baseline --no-solve --no-pmc --sustain-s 5 > $O/bench_w30_one_rank_rccl.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-600 $O/bench_w30_one_rank_rccl.json
