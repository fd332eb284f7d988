#!/bin/bash
# round 5, GPU call 8k: the closing round on HEAD (after the multiset overflow set and the 36 * 2^30-point table) -- plain bench / the same under rocprofv3 --kernel-trace --stats / one PMC group per pass (never combined with a trace);
# the driver's own command (python bench.py, every leg); smoke(); the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08k; mkdir -p $O; cd $R
export TMPDIR=/tmp
git -C $R rev-parse HEAD > $O/head.txt 2>/dev/null
bash tools/profile_round.sh r08k 2>&1 | tail -40 | tee $O/profile_round.log
cd $R
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $O/smoke.log
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_gpu.log
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
