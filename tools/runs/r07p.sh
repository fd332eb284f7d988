#!/bin/bash
# round 5, GPU call 7p: the profile round of the shipped binary (after the <4, ., .> instantiation and the placement rule for lines above 0.6 of the HBM) -- plain bench / the same
# under rocprofv3 --kernel-trace --stats / one PMC group per pass (never combined with a trace) at -w 30; the kernel trace of the 64-byte any-bucket kernel at -w 35; the driver's
# own command (python bench.py, every leg); and the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07p; mkdir -p $O; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r07p 2>&1 | tail -40 | tee $O/profile_round.log
cd /tmp; rm -rf /tmp/rp35
( rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp35 -- python $R/bench.py --w 35 --htsz 3221225472 --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/bench_w35_under_rocprofv3_stats.json 2> /tmp/rp35.err; echo "w35 trace rc $?"
  python $R/tools/rocprof_summary.py stats /tmp/rp35 $O/rocprofv3_kernel_stats_w35.csv | head -8 ) 2>&1 | tee $O/w35_trace.log
cd $R
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_gpu.log
