#!/bin/bash
# round 5, GPU call 7b: census + sampled membership (small tables in every layout, -w 34), a first -w 35 table with 1.5 * 2^30 buckets of 128 bytes, and twelve more
# alternating processes round-4 library / new library (r07a's eight said -2 % for identical hot loops: noise or the missing scratch segment?)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07b; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "census or w34" 2>&1 | tail -12 ) | tee $O/pytest_census.log
( BSGS_BUILD_VERBOSE=1 timeout 900 python bench.py --w 35 --htsz 1610612736 --no-pmc --no-solve --no-cpu-baseline --sustain-s 5 > $O/bench_w35.json 2> $O/bench_w35.err; echo "w35 rc $?"; tail -5 $O/bench_w35.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w35.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("w35 value %.2f G ms/launch %.3f tpl %d build %s setup %.1f fp hits %d eff keys/s %.3e" % (d["value"]/1e9, r["avg_launch_ms"], r["tiles_per_launch"], d["table_build"], d["setup_s"], d["false_positive_hits"], d["effective_keys_per_s"]))
    print(d["chain_scratch"], d["config"]["workload"])
except Exception as e: print("w35 FAILED", e)
PY
) 2>&1 | tee $O/w35.log
B=$R/bsgs-cuda_amd/build
one() { env $1 timeout 300 python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}; c=d['chain_scratch']
print('$2  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz  %.0f W  grades %.1f..%.1f sep %s' % (d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), p.get('socket_W_mean',0), c.get('worst_kept_grade_G_per_s',0), c.get('best_grade_G_per_s',0), c.get('separated')))"; }
for v in A B B A A B B A A B B A; do
  if [ $v = A ]; then one "BSGS_LIB_PATH=$B/exp_r04/libbsgs_hip.so" A; else one "BSGS_LIB_PATH=$B/libbsgs_hip.so" B; fi
done 2>&1 | tee $O/abba12_r04_vs_new.log
