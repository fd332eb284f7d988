#!/bin/bash
# tools/profile_round.sh <tag> [extra bench args] -- run on the GPU box (through gpurun); writes summaries to gpurun_out/prof_<tag>/
# pass 1: kernel trace + stats of the default bench; further passes: one PMC group each (never combined with traces).
# All passes run with --tune-candidates 1: the start-up placement tuning launches the same kernel on candidate placements that are
# then discarded, and their (slower) launches would be averaged into the per-kernel statistics the bench line is compared with.
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp && mkdir -p /tmp/rp
# the plain line first, then the same command under the kernel trace: a pair of processes on one box (their launch times must agree)
python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --tune-candidates 1 "$@" > $OUT/bench_plain.json 2> /tmp/rp/plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp/stats -- python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --tune-candidates 1 "$@" > $OUT/bench_under_rocprofv3_stats.json 2> /tmp/rp/stats.err
python $R/tools/rocprof_summary.py stats /tmp/rp/stats $OUT/rocprofv3_kernel_stats.csv > /dev/null
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32" \
           "VALUBusy" "SALUBusy" "MemUnitStalled" "MeanOccupancyPerCU" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d /tmp/rp/pmc$i -- python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 --warmup-s 0 --tune-candidates 1 --steps 3 --warmup 1 "$@" > $OUT/bench_under_pmc_$name.json 2> /tmp/rp/pmc$i.err \
    && python $R/tools/rocprof_summary.py pmc /tmp/rp/pmc$i $OUT/rocprofv3_pmc_$name.csv > /dev/null || { echo "pass $name failed"; tail -3 /tmp/rp/pmc$i.err; }
done
grep -h "giant_\|gups" $OUT/rocprofv3_pmc_*.csv | sed 's/^"[^"]*",//' | sort | uniq | head -80
head -5 $OUT/rocprofv3_kernel_stats.csv | cut -c1-200
for f in $OUT/bench_plain.json $OUT/bench_under_rocprofv3_stats.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.2f G' % (d['value']/1e9), '%.3f ms/launch' % d['roofline']['avg_launch_ms'], 'sustained %.2f G' % ((d.get('value_sustained') or 0)/1e9))"; done
