#!/bin/bash
# round 3, GPU call U: cache-policy bits of the probe line loads (global_load_lds ... sc0/sc1/nt): default 2 (nt) against 3 (sc0 nt), 16 (sc1), 18 (sc1 nt), 19 (sc0 sc1 nt);
# and the bench flags no test exercises (--centres host, --table synthetic, --tune-candidates 3)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
for c in 3 16 18 19; do
  echo "## A = cpol 2 (shipped), B = cpol $c"
  STEPS=20 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip.so" "BSGS_LIB_PATH=$B/libbsgs_hip_cpol$c.so"
done > $O/abba_probe_cpol.log 2>&1
cat $O/abba_probe_cpol.log
for extra in "--centres host --steps 4 --warmup 1" "--table synthetic --w 26 --htsz 25" "--tune-candidates 3 --w 26 --htsz 25"; do
  python bench.py $extra --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 2> $O/flags.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$extra', '%.2f G' % (d['value']/1e9), d['config']['centres'], d['placement_tuning'])" || tail -3 $O/flags.err
done
