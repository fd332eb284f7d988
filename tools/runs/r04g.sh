#!/bin/bash
# round 3, GPU call 4g: the grader's stop rule with both classes seen: eight consecutive bench processes at -w 30, two at -w 26, with the grades printed
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04g; mkdir -p $O; cd $R
export TMPDIR=/tmp
for cfg in "30 28" "30 28" "30 28" "30 28" "30 28" "30 28" "30 28" "30 28" "26 25" "26 25"; do set -- $cfg
  python bench.py --w $1 --htsz $2 --no-cpu-baseline --no-pmc --no-solve --sustain-s 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['chain_scratch']
print('-w $1  %.2f G  sustained %.2f G  %.3f ms/launch  graded %d kept %d  grades %.1f..%.1f  setup %.1f s' % (d['value']/1e9, d['value_sustained']/1e9, d['roofline']['avg_launch_ms'], c['graded'], c['pieces'], c['worst_kept_grade_G_per_s'], c['best_grade_G_per_s'], d['setup_s']))"
done | tee $O/repeat_runs_after_grader_fix.log
