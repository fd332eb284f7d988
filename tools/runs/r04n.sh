#!/bin/bash
# round 3, GPU call 4n: what a ONE-tile launch (the reference's own launch pattern, route A without prediction) could reach with another internal batching
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04n; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { # label, env, flags
  out=$(env $2 python bench.py --w 26 --htsz 25 $3 --steps 300 --warmup 30 --warmup-s 1 --sustain-s 0 --no-solve --no-pmc --no-cpu-baseline 2>$O/err.log | tail -1)
  python - "$1" "$out" <<'PY'
import json,sys
try:
    d=json.loads(sys.argv[2]); print("%-44s %6.2f G  %8.3f ms/launch  tiles/launch %d  kernel %s" % (sys.argv[1], d['value']/1e9, d['ms_per_step'], d['config']['tiles_per_step'], d['roofline']['kernel']))
except Exception as e: print(sys.argv[1], "FAILED", e, sys.argv[2][:300])
PY
}
{
run "tpl 1, 16384 x 1024 (default batching)" "A=1" "--tiles-per-launch 1"
run "tpl 1, 65536 x 256 (BATCH_MULT=1)" "BSGS_BATCH_MULT=1" "--tiles-per-launch 1"
run "tpl 1, -t 512 -p 128: 131072 x 128" "BSGS_BATCH_MULT=1" "--tiles-per-launch 1 -t 512 -b 256 -p 128"
run "tpl 1, -t 512 -b 512 -p 64: 262144 x 64" "BSGS_BATCH_MULT=1" "--tiles-per-launch 1 -t 512 -b 512 -p 64"
run "tpl 2, default batching" "A=1" "--tiles-per-launch 2"
run "tpl 4, default batching" "A=1" "--tiles-per-launch 4"
run "tpl 4, 65536 x 256" "BSGS_BATCH_MULT=1" "--tiles-per-launch 4"
run "tpl 16, default batching" "A=1" "--tiles-per-launch 16"
} | tee $O/one_tile_launch_batching.log
tail -3 $O/err.log
