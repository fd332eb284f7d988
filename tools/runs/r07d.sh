#!/bin/bash
# round 5, GPU call 7d: the GPU suite again (r07c stopped at a test's own string count), then -w 35 with the 128-byte-line kernel in two-wave blocks (ten waves per CU instead
# of eight, 135 VGPRs, no spills) against four-wave blocks (BSGS_LINES128_BLOCK=256), and the unsaturated-limb multipliers (exactness, then sustained rate and power)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07d; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) | tee $O/pytest_gpu.log
for blk in 128 256; do
  ( BSGS_LINES128_BLOCK=$blk timeout 900 python bench.py --w 35 --htsz 1610612736 --no-pmc --no-solve --no-cpu-baseline --sustain-s 5 > $O/bench_w35_block$blk.json 2> $O/bench_w35_block$blk.err; echo "w35 block $blk rc $?"; tail -2 $O/bench_w35_block$blk.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w35_block$blk.json").read().strip().splitlines()[-1]); r=d["roofline"]; p=d["alu"]["power"] or {}
    print("w35 block $blk: value %.2f G sustained %.2f ms/launch %.3f tpl %d sclk %.0f W %.0f fp hits %d eff keys/s %.3e kernel %s" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], p.get("sclk_MHz_mean",0), p.get("socket_W_mean",0), d["false_positive_hits"], d["effective_keys_per_s"], r["kernel"]))
except Exception as e: print("w35 FAILED", e)
PY
  ) 2>&1 | tee -a $O/w35_blocks.log
done
MB=$R/bsgs-cuda_amd/build/microbench
( $MB unsatcheck; $MB dpfcheck > /dev/null; OPS="200 203 204 201" bash tools/power_ops.sh ) 2>&1 | tee $O/modmul_variants.log
