#!/bin/bash
# round 3, GPU call O: where the waves wait at -w 34 (SQ wait counters) next to -w 30
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03o; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in "30 28" "34 31"; do set -- $cfg
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM"; do
    name=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/rk; rocprofv3 --pmc $grp --output-format csv -d /tmp/rk -- python $R/bench.py --w $1 --htsz $2 --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 --warmup-s 0 --steps 3 --warmup 1 > $O/bench_w$1_$name.json 2> /tmp/rk.err \
      && python $R/tools/rocprof_summary.py pmc /tmp/rk $O/pmc_w$1_$name.csv > /dev/null || { echo "pass w$1 $name failed"; tail -3 /tmp/rk.err; }
  done
done
for w in 30 34; do echo "## w$w"; grep -h "false, false" $O/pmc_w${w}_*.csv | sed 's/^"[^"]*",//'; done
for f in $O/bench_w*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], (d['alu']['power'] or {}).get('sclk_MHz_mean'))"; done
