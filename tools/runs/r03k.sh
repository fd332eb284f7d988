#!/bin/bash
# round 3, GPU call K: why -w 34 runs at 38 G with the clock at 1.85 GHz: L1->L2 read latency and translation counters at -w 30 and -w 34
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -iE "UTCL|TLB|translat" | cut -c1-160 | head -40 > $O/counters_translation.txt
for cfg in "30 28" "34 31"; do set -- $cfg
  for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
    name=$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/rk; rocprofv3 --pmc $grp --output-format csv -d /tmp/rk -- python $R/bench.py --w $1 --htsz $2 --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 --warmup-s 0 --steps 3 --warmup 1 > $O/bench_w$1_$name.json 2> /tmp/rk.err \
      && python $R/tools/rocprof_summary.py pmc /tmp/rk $O/pmc_w$1_$name.csv > /dev/null || { echo "pass w$1 $name failed"; tail -3 /tmp/rk.err; }
  done
done
cat $O/counters_translation.txt | head -20
grep -h "false, false" $O/pmc_w*.csv | sed 's/^"[^"]*",//'
for f in $O/bench_w*.json; do python -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), '%.2f ms' % d['roofline']['avg_launch_ms'], (d['alu']['power'] or {}).get('sclk_MHz_mean'))"; done
