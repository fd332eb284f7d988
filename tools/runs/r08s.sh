#!/bin/bash
# round 5, GPU call 8s: is the ONE counter of the overflow list what holds the scatter of the large tables above its random-write bound?  The builder's stage clocks at 36 * 2^30
# and 2^35 points with the shipped library and with a diagnostic copy whose generator drops the arrivals beyond a line's slots instead of appending them (build/exp_nolist: the
# table it leaves is WRONG and the run ends at its validation; only the "generate + scatter" stage is read)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08s; mkdir -p $O; cd $R
export TMPDIR=/tmp
for lib in build build/exp_nolist build; do
  for w in 38654705664 35; do
    echo "== $lib  w $w"
    BSGS_LIB_PATH=$R/bsgs-cuda_amd/$lib/libbsgs_hip.so BSGS_BUILD_VERBOSE=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 2 --warmup 1 --warmup-s 0 --w $w --htsz 3221225472 --layout 4 2>&1 | grep -E "^\[build\]|rror" | head -8
  done
done 2>&1 | tee $O/scatter_with_and_without_list_appends.log
