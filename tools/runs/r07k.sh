#!/bin/bash
# round 5, GPU call 7k: what the 64-byte-line kernel does on FULLER lines (would 3 * 2^30 lines of 64 bytes carry -w 35 better than 1.5 * 2^30 lines of 128 bytes?).
# 2^31 lines of 64 bytes (--layout 4) at mean load 8 (-w 34), 10.67 (-w 34.415 = 4/3 * 2^34: -w 35's load on 3 * 2^30 lines), 12 (-w 34.585): giant-steps/s, over-full lines.
# (the first two attempts of this call let bench.py choose the layout: it took 128-byte lines for w > 2^34 -- 256 GiB, 48-tile launches -- and measured something else)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07k; mkdir -p $O; cd $R
export TMPDIR=/tmp
for W in 34 34.415 34.585; do for GL in 1 0; do
  [ $GL = 0 ] && [ -s $O/bench_w${W}_lines64.json ] && continue
  BSGS_GRADED_LINES=$GL python bench.py --w $W --htsz 31 --layout 4 --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/bench_w${W}_lines64.json 2> $O/bench_w${W}_gl$GL.err
  python - $O/bench_w${W}_lines64.json $GL <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], 'graded_lines', sys.argv[2], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "keys/s %.3e" % d["effective_keys_per_s"], "tiles per launch", d["roofline"]["tiles_per_launch"],
      d["config"]["table_layout"], "over-full", d["config"]["overflow_buckets"], "build %.2f s" % d["table_build"]["seconds"], "hits", d.get("false_positive_hits"), "scratch", d.get("chain_scratch", {}).get("from_reserved_group"))
P
done; done 2>&1 | tee $O/fuller_lines.log
