#!/bin/bash
# round 4, GPU call 6p: cache policy of the probe line loads (LDS-DMA): shipped nt (2) against nt+sc1 (18), sc0+sc1+nt (19), sc0+sc1 (17), sc0+nt (3).
# Exact results in every variant (a cache policy changes no value); A B B A per variant, then the full ABBA for anything that wins
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
one() { env $1 python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('$2  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz  [%s] fp hits %d' % (d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['library_build_info'], d['false_positive_hits']))"; }
for c in 18 19 17 3; do
  echo "# probe cache policy $c against the shipped 2"
  one "BSGS_LIB_PATH=$B/libbsgs_hip.so" A; one "BSGS_LIB_PATH=$B/exp_cpol$c/libbsgs_hip.so" B; one "BSGS_LIB_PATH=$B/exp_cpol$c/libbsgs_hip.so" B; one "BSGS_LIB_PATH=$B/libbsgs_hip.so" A
done 2>&1 | tee $O/probe_cache_policy.log
