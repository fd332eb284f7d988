#!/bin/bash
# round 3, GPU call T: six consecutive bench processes on one box (run-to-run spread of the final code), at -w 30 and -w 34
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
export TMPDIR=/tmp
for w in "30 28" "34 31"; do set -- $w
  for i in 1 2 3 4 5 6; do
    python bench.py --w $1 --htsz $2 --no-cpu-baseline --no-pmc --no-solve --sustain-s 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('-w $1 run $i  %.2f G  sustained %.2f G  %.3f ms/launch  sclk %.0f MHz  %s' % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['chain_scratch']['from_reserved_group']))"
  done
done | tee $O/repeat_runs.log
