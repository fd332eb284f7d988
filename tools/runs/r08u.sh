#!/bin/bash
# round 5, GPU call 8u: the closing round on the FINAL HEAD (the builder's overflow list by regions came after r08k): smoke(), the whole GPU suite, the driver's own command
# (python bench.py, every leg), the same under rocprofv3 --kernel-trace --stats, config 3 at -w auto
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08u; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) | tee $O/smoke.log
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 ) | tee $O/pytest_gpu.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
cd /tmp && rm -rf /tmp/rpu && mkdir -p /tmp/rpu
python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --tune-candidates 1 > $O/bench_plain.json 2> /tmp/rpu/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rpu/stats -- python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --tune-candidates 1 > $O/bench_under_rocprofv3_stats.json 2> /tmp/rpu/stats.err
python $R/tools/rocprof_summary.py stats /tmp/rpu/stats $O/rocprofv3_kernel_stats.csv > /dev/null
head -3 $O/rocprofv3_kernel_stats.csv | cut -c1-200
cd $R
( python tools/config3_run.py 0.5 /tmp/cfg3u "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
