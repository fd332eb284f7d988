#!/bin/bash
# round 5, GPU call 7z: is it the overflow SET next to the scratch that makes the reserve lose at 192 GiB?  3 * 2^30 lines of 64 bytes at load 8 (-w 34.585: a set of < 1 GiB)
# with the reserve forced (BSGS_RESERVE_ANYWAY=1) and without, alternating; then -w 35 (set 16 GiB) with the reserve once more on this box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07z; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { local name=$1 envs=$2; shift 2
  env $envs BSGS_BUILD_VERBOSE=1 timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["alu"]["power"]
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "alloc %.2f s" % d["table_build"]["allocation_and_placement_seconds"],
          "sclk %.0f MHz" % p["sclk_MHz_mean"], "socket %.0f W" % p["socket_W_mean"], "over-full", d["config"]["overflow_buckets"], "scratch from reserve", d.get("chain_scratch", {}).get("from_reserved_group"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  grep "^\[place\]" $O/$name.err | cut -c1-220
}
( run load8_192GiB_reserved_1 BSGS_RESERVE_ANYWAY=1 --w 34.585 --htsz 3221225472 --layout 4
  run load8_192GiB_plain_1 X=1 --w 34.585 --htsz 3221225472 --layout 4
  run load8_192GiB_plain_2 X=1 --w 34.585 --htsz 3221225472 --layout 4
  run load8_192GiB_reserved_2 BSGS_RESERVE_ANYWAY=1 --w 34.585 --htsz 3221225472 --layout 4
  run w35_192GiB_reserved BSGS_RESERVE_ANYWAY=1 --w 35 --htsz 3221225472 --layout 4 ) 2>&1 | tee $O/reserve_at_192GiB_small_set.log
