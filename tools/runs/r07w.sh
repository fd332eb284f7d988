#!/bin/bash
# round 5, GPU call 7w: -w 35 (3 * 2^30 lines of 64 bytes, 208 GiB of table) with smaller launches = less chain scratch (24 GiB at 192 tiles): does the 8 % the 192 GiB footprint
# costs (r07m) come from the scratch streams sharing memory groups with the lines?
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07w; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 600 python bench.py --w 35 --htsz 3221225472 --layout 4 "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["alu"]["power"]
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "tiles per launch", d["roofline"]["tiles_per_launch"], "ms/launch %.2f" % d["ms_per_step"],
          "sclk %.0f MHz" % p["sclk_MHz_mean"], "socket %.0f W" % p["socket_W_mean"], "big buffers", d.get("big_buffers_GiB"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
( run tpl192_1; run tpl64 --tiles-per-launch 64; run tpl128 --tiles-per-launch 128; run tpl96 --tiles-per-launch 96; run tpl192_2 ) 2>&1 | tee $O/w35_tiles_per_launch.log
