#!/bin/bash
# round 5, GPU call 7s: the tile kernel when its probes are CACHE-resident -- the same code on small tables: 2^22 bucket lines (256 MiB: the memory-side cache), 2^21
# (128 MiB), 2^18 (16 MiB: the L2s), against the headline table (2^28 lines, 16 GiB: HBM), one box.  The ceiling of any scheme that sorts the probes by table region.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07s; mkdir -p $O; cd $R
export TMPDIR=/tmp
run() { local name=$1; shift
  timeout 600 python bench.py "$@" --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 8 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = d["alu"]["power"]
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "ms/launch %.2f" % d["ms_per_step"], "sclk %.0f MHz" % p["sclk_MHz_mean"], "socket %.0f W" % p["socket_W_mean"],
          "nJ/step", d.get("nJ_per_giant_step"), "hits", d.get("false_positive_hits"), d["config"]["table_layout"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
}
( run w30_htsz28_hbm_1
  run w24_htsz22_256MiB --w 24 --htsz 22
  run w23_htsz21_128MiB --w 23 --htsz 21
  run w20_htsz18_16MiB --w 20 --htsz 18
  run w26_htsz24_1GiB --w 26 --htsz 24
  run w30_htsz28_hbm_2 ) 2>&1 | tee $O/cache_resident_tables.log
