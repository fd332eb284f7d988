#!/bin/bash
# round 5, GPU call 7n: -w 35 on ONE box, with and without the memory group reserved for the chain scratch (r07m: the 64-byte-line table that lost in r07l, 28.5 G, runs at
# 36.3 G when no group is reserved): 1.5 * 2^30 lines of 128 bytes (shipped) and 3 * 2^30 lines of 64 bytes, each both ways, A B B A.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07n; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs BSGS_BUILD_VERBOSE=1 timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "keys/s %.3e" % d["effective_keys_per_s"], "tiles per launch", d["roofline"]["tiles_per_launch"],
          d["config"]["table_layout"], "over-full", d["config"]["overflow_buckets"], "build %.2f s" % d["table_build"]["seconds"], "alloc %.2f s" % d["table_build"]["allocation_and_placement_seconds"],
          "hits", d.get("false_positive_hits"), "scratch", d.get("chain_scratch", {}).get("from_reserved_group"), d["roofline"]["kernel"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  grep "^\[place\]" $O/$name.err | cut -c1-300
}
( run w35_lines128_reserved_1 X=1 --w 35 --htsz 1610612736 --layout 5
  run w35_lines128_plain_1 BSGS_GRADED_LINES=0 --w 35 --htsz 1610612736 --layout 5
  run w35_lines64_plain_1 BSGS_GRADED_LINES=0 --w 35 --htsz 3221225472 --layout 4
  run w35_lines64_reserved_1 X=1 --w 35 --htsz 3221225472 --layout 4
  run w35_lines64_plain_2 BSGS_GRADED_LINES=0 --w 35 --htsz 3221225472 --layout 4
  run w35_lines128_plain_2 BSGS_GRADED_LINES=0 --w 35 --htsz 1610612736 --layout 5
  run w35_lines128_reserved_2 X=1 --w 35 --htsz 1610612736 --layout 5 ) 2>&1 | tee $O/w35_reserved_group_or_not.log
