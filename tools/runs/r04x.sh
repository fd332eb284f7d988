#!/bin/bash
# round 3, GPU call 4x: one Fermat inversion per block (B, the new default) against one per wave (A = -DBSGS_INV_PER_WAVE): parity, ABBA at the headline size, small launches
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04x; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
( timeout 1200 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -6 ) > $O/pytest.log; cat $O/pytest.log
{ echo "# A = one inversion per wave (libbsgs_hip_invwave.so), B = one per block (shipped)"; STEPS=30 bash tools/abba.sh "BSGS_LIB_PATH=$B/libbsgs_hip_invwave.so" "BSGS_LIB_PATH=$B/libbsgs_hip.so"; } 2>&1 | tee $O/abba_inversion_per_block.log
run() { # label, env, flags
  out=$(env $2 python bench.py --w 26 --htsz 25 $3 --steps 300 --warmup 30 --warmup-s 1 --sustain-s 0 --no-solve --no-pmc --no-cpu-baseline 2>$O/err.log | tail -1)
  python - "$1" "$out" <<'PY'
import json,sys
try:
    d=json.loads(sys.argv[2]); print("%-50s %6.2f G  %8.3f ms/launch" % (sys.argv[1], d['value']/1e9, d['ms_per_step']))
except Exception as e: print(sys.argv[1], "FAILED", e, sys.argv[2][:300])
PY
}
{
for n in 1 2 4 8; do
run "tiles per launch $n, inversion per wave" "BSGS_LIB_PATH=$B/libbsgs_hip_invwave.so" "--tiles-per-launch $n"
run "tiles per launch $n, inversion per block" "A=1" "--tiles-per-launch $n"
done
run "1 tile, per block, -t 512 -b 512 -p 64 forced (262144 x 64)" "BSGS_BATCH_MULT=1 BSGS_NARROW_LAUNCHES=0" "--tiles-per-launch 1 -t 512 -b 512 -p 64"
run "1 tile, per block, -t 1024 -b 512 -p 32 forced (524288 x 32)" "BSGS_BATCH_MULT=1 BSGS_NARROW_LAUNCHES=0" "--tiles-per-launch 1 -t 1024 -b 512 -p 32"
run "2 tiles, per block, -t 512 -b 512 -p 64 forced (262144 x 64)" "BSGS_BATCH_MULT=1 BSGS_NARROW_LAUNCHES=0" "--tiles-per-launch 2 -t 512 -b 512 -p 64"
} | tee $O/small_launches_inversion_per_block.log
