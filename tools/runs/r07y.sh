#!/bin/bash
# round 5, GPU call 7y: the boundary of the placement rule (lines above 0.6 of the HBM: no reserved memory group) -- -w 34 -htsz 31 (128 GiB of lines: reserve kept) with and
# without the reserve, alternating, and -w 35 on 2.75 * 2^30 lines (176 GiB: now plain; with the reserve 32.9 G in r07l), one box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07y; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { local name=$1 envs=$2; shift 2
  env $envs BSGS_BUILD_VERBOSE=1 timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["alu"]["power"]
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "alloc %.2f s" % d["table_build"]["allocation_and_placement_seconds"],
          "sclk %.0f MHz" % p["sclk_MHz_mean"], "socket %.0f W" % p["socket_W_mean"], "scratch from reserve", d.get("chain_scratch", {}).get("from_reserved_group"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  grep "^\[place\]" $O/$name.err | cut -c1-200
}
( run w34_reserved_1 X=1 --w 34 --htsz 31 --layout 4
  run w34_plain_1 BSGS_GRADED_LINES=0 --w 34 --htsz 31 --layout 4
  run w34_plain_2 BSGS_GRADED_LINES=0 --w 34 --htsz 31 --layout 4
  run w34_reserved_2 X=1 --w 34 --htsz 31 --layout 4
  run w35_176GiB_plain X=1 --w 35 --htsz 2952790016 --layout 4 ) 2>&1 | tee $O/placement_rule_boundary.log
