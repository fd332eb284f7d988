#!/bin/bash
# round 6, GPU call e: the whole GPU suite (overflow-list tail, checkpoint at job end), the driver's command, and the round's profile pair + PMC passes (tools/profile_round.sh)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10e; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
( timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) | tee $O/pytest_gpu.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
bash tools/profile_round.sh r10e 2>&1 | tail -40 | tee $O/profile_round.log
