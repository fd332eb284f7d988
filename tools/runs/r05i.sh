#!/bin/bash
# round 3, GPU call 5i: launches sized by WORK (192 x 2^24 giants) whatever the geometry: GPU suite, the README examples again, the headline (unchanged: 192 tiles)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05i; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
run() {
  out=$(python bench.py $1 --steps 10 --warmup 3 --sustain-s 5 --no-solve --no-pmc --no-cpu-baseline 2>$O/err.log | tail -1)
  python - "$1" "$out" <<'PY'
import json,sys
try:
    d=json.loads(sys.argv[2]); print("%-50s %6.2f G  sustained %6.2f G  %8.3f ms/launch  tiles/launch %d  %s" % (sys.argv[1], d['value']/1e9, d['value_sustained']/1e9, d['ms_per_step'], d['config']['tiles_per_step'], d['roofline']['kernel']))
except Exception as e: print(sys.argv[1], "FAILED", e, sys.argv[2][:300])
PY
}
{
run "-t 256 -b 88 -p 130 --w 29.87 --htsz 28"
run "-t 256 -b 272 -p 220 --w 30.5 --htsz 29"
run "-t 256 -b 138 -p 244 --w 30.25 --htsz 28"
run "-t 256 -b 256 -p 256 --w 30 --htsz 28"
} | tee $O/reference_readme_examples.log
