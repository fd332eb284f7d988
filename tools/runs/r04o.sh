#!/bin/bash
# round 3, GPU call 4o: small launches on a narrow batching (pick_batching): new tests, whole GPU suite, launch-size sweep against BSGS_NARROW_LAUNCHES=0
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "small_launches or whole_tile" 2>&1 | tail -15 ) > $O/pytest_new.log; cat $O/pytest_new.log
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
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
for n in 1 2 3 4 6 8 12 16 24; do
run "tiles per launch $n, default batching only" "BSGS_NARROW_LAUNCHES=0" "--tiles-per-launch $n"
run "tiles per launch $n, narrow batching" "A=1" "--tiles-per-launch $n"
done
} | tee $O/launch_size_sweep.log
cp gpurun_out/route_a_throughput.json $O/ 2>/dev/null
