#!/bin/bash
# round 3, GPU call 5h: the three example command lines of the reference's README (README.md:28-34) through bench.py: rate, kernel, batching
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() {
  out=$(python bench.py $1 --steps 10 --warmup 3 --sustain-s 5 --no-solve --no-pmc --no-cpu-baseline 2>$O/err.log | tail -1)
  python - "$1" "$out" <<'PY'
import json,sys
try:
    d=json.loads(sys.argv[2]); print("%-50s %6.2f G  sustained %6.2f G  %8.3f ms/launch  tiles/launch %d  %s  %s" % (sys.argv[1], d['value']/1e9, d['value_sustained']/1e9, d['ms_per_step'], d['config']['tiles_per_step'], d['roofline']['kernel'], d['config']['table_layout']))
except Exception as e: print(sys.argv[1], "FAILED", e, sys.argv[2][:300])
PY
}
{
run "-t 256 -b 88 -p 130 --w 29.87 --htsz 28"
run "-t 256 -b 272 -p 220 --w 30.5 --htsz 29"
run "-t 256 -b 138 -p 244 --w 30.25 --htsz 28"
} | tee $O/reference_readme_examples.log
tail -2 $O/err.log
