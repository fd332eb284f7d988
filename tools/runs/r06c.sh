#!/bin/bash
# round 4, GPU call 6c: what bounds the table builder (row f1): scatter microbench (atomics by scope, XCD-partitioned) and rocprofv3 kernel stats of the builder at -w 30 and -w 34
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
export TMPDIR=/tmp
$R/bsgs-cuda_amd/build/microbench scatter 16384 131072 2>&1 | tee $O/scatter_microbench.jsonl
BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 30 28 2>&1 | tee $O/build_w30.log
BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 34 31 2>&1 | tee $O/build_w34.log
cd /tmp
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w30 -- python $R/tools/build_bench.py 30 28 > $O/prof_w30.log 2>&1
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w34 -- python $R/tools/build_bench.py 34 31 > $O/prof_w34.log 2>&1
cd $R
for w in w30 w34; do f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/builder_${w}_kernel_stats.csv; rm -rf $O/prof_$w; done
head -12 $O/builder_w30_kernel_stats.csv; head -12 $O/builder_w34_kernel_stats.csv
