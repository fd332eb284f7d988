#!/bin/bash
# round 5, GPU call 7t: the host's large images first-touched by 16 threads on huge pages (HostBuf::resize) -- the "table + giants files (load, or build + save)" stage of a
# -w 30 -htsz 28 start-up (the file writers now start after the engines hold their tables) with and without (BSGS_HOST_NO_PREFAULT=1), building and loading, alternating; then BASELINE config 4 (1000 keys) with it
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07t; mkdir -p $O; cd $R
export TMPDIR=/tmp
H=$R/bsgs-cuda_amd/build/bsgs_mi355x
one() { # name env
  local d=/tmp/r07t_$1; rm -rf $d; mkdir -p $d
  for what in build load; do
    local t0=$(date +%s.%N)
    env $2 $H -onlygen -dir $d -t 256 -b 256 -p 256 -w 30 -htsz 28 > $d/out.txt 2> $d/err.txt
    local t1=$(date +%s.%N)
    echo "$1 $what: $(grep 'table + giants files' $d/out.txt | sed 's/  */ /g') | $(grep 'Done in' $d/out.txt) | process $(python3 -c "print('%.2f' % ($t1 - $t0))") s"
  done
  rm -rf $d
}
( one prefault_1 X=1; one plain_1 BSGS_HOST_NO_PREFAULT=1; one prefault_2 X=1; one plain_2 BSGS_HOST_NO_PREFAULT=1 ) 2>&1 | tee $O/host_images_prefault.log
( python tools/config4_run.py 1000 /tmp/cfg4t ) 2>&1 | tee $O/config4_1000keys_prefault.json
( BSGS_SHORT_JOB_BATCHES=18 python tools/config4_run.py 1000 /tmp/cfg4u "-lanes 3" ) 2>&1 | tee $O/config4_1000keys_prefault_three_lanes_18_batches.json
