#!/bin/bash
# round 4, GPU call 6v: rocprofv3 kernel stats of the REBUILT table builder at -w 30 (reference-format image) and -w 34 (extended table)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06v; mkdir -p $O; cd /tmp
export TMPDIR=/tmp
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w30 -- python $R/tools/build_bench.py 30 28 > $O/prof_w30.log 2>&1
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_w34 -- python $R/tools/build_bench.py 34 31 > $O/prof_w34.log 2>&1
cd $R
for w in w30 w34; do f=$(find $O/prof_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python - "$f" $O/builder_${w}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[(r[0][:140] + " ...") if len(r[0]) > 140 else r[0]] + r[1:] for r in rows[1:]]
csv.writer(open(sys.argv[2], "w", newline="")).writerows(out)
for r in out[:9]: print(",".join(r)[:230])
PY
rm -rf $O/prof_$w; done
grep "^{" $O/prof_w30.log $O/prof_w34.log
