#!/bin/bash
# round 5, GPU call 7i: the profile round of the shipped binary -- plain bench / the same under rocprofv3 --kernel-trace --stats / one PMC group per pass (never combined
# with a trace) at -w 30; the kernel trace of the 128-byte-line kernel at -w 35; and the whole GPU suite once more
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07i; mkdir -p $O; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r07i 2>&1 | tail -40 | tee $O/profile_round.log
cd /tmp; rm -rf /tmp/rp35
( rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp35 -- python $R/bench.py --w 35 --htsz 1610612736 --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/bench_w35_under_rocprofv3_stats.json 2> /tmp/rp35.err; echo "w35 trace rc $?"
  python $R/tools/rocprof_summary.py stats /tmp/rp35 $O/rocprofv3_kernel_stats_w35.csv | head -8 ) 2>&1 | tee $O/w35_trace.log
cd $R
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) | tee $O/pytest_gpu.log
