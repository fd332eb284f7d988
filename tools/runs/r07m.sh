#!/bin/bash
# round 5, GPU call 7m: (1) r07k again (its records were lost with the container): the 64-byte-line kernel on FULLER lines -- 2^31 lines of 64 bytes (128 GiB) at mean load 8
# (-w 34), 10.67 (-w 34.415), 12 (-w 34.585); (2) why -w 35 on 3 * 2^30 lines of 64 bytes loses (r07l: 28.5 G): the same 192 GiB of lines at load 8 (-w 34.585 -htsz 3221225472:
# hardly any overflow probes) separates the footprint from the overflow path; the same table without the reserved memory group (BSGS_GRADED_LINES=0) separates the placement
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07m; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { # name, env, args...
  local name=$1 envs=$2; shift 2
  env $envs BSGS_BUILD_VERBOSE=1 timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "keys/s %.3e" % d["effective_keys_per_s"], "tiles per launch", d["roofline"]["tiles_per_launch"],
          d["config"]["table_layout"], "over-full", d["config"]["overflow_buckets"], "build %.2f s" % d["table_build"]["seconds"], "hits", d.get("false_positive_hits"), "scratch", d.get("chain_scratch", {}).get("from_reserved_group"),
          d["roofline"]["kernel"], "clock", d.get("clock"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
( for W in 34 34.415 34.585; do run bench_w${W}_2e31_lines64 X=1 --w $W --htsz 31 --layout 4; done ) 2>&1 | tee $O/fuller_lines.log
( run bench_w34.585_3x2e30_lines64 X=1 --w 34.585 --htsz 3221225472 --layout 4
  run bench_w35_3x2e30_lines64_no_reserved_group BSGS_GRADED_LINES=0 --w 35 --htsz 3221225472 --layout 4
  run bench_w34.585_1.5x2e30_lines128 X=1 --w 34.585 --htsz 1610612736 --layout 5 ) 2>&1 | tee $O/w35_lines64_diagnosis.log
