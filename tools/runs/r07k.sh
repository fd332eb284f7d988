#!/bin/bash
# round 5, GPU call 7k: what the 64-byte-line kernel does on FULLER lines (would 3 * 2^30 lines of 64 bytes carry -w 35 better than 1.5 * 2^30 lines of 128 bytes?).
# 2^31 lines of 64 bytes at mean load 8 (-w 34), 10.67 (-w 34.415 = 4/3 * 2^34: -w 35's load on 3 * 2^30 lines), 12 (-w 34.585): giant-steps/s and over-full lines
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07k; mkdir -p $O; cd $R
export TMPDIR=/tmp
for W in 34 34.415 34.585; do
  python bench.py --w $W --htsz 31 --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/bench_w${W}_htsz31.json 2> $O/bench_w${W}.err
  python - $O/bench_w${W}_htsz31.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "keys/s %.3e" % d["effective_keys_per_s"], "table", d.get("table_build"), "hits", d.get("false_positive_hits"))
P
done 2>&1 | tee $O/fuller_lines.log
