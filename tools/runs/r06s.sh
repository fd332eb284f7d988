#!/bin/bash
# round 4, GPU call 6s: the slice gate (exact results: -DBSGS_SLICE_GATE=W holds a block back while it is more than W rows of giants ahead of the slowest running block
# of its (chunk, slice) group, so that the 64 blocks that walk one slice share it through the XCD's L2): parity on the gated library, then A B B A per W, FETCH_SIZE with and without
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
export TMPDIR=/tmp
B=$R/bsgs-cuda_amd/build
( BSGS_LIB_PATH=$B/exp_gate64/libbsgs_hip.so timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "not bench and not abi and not variants" 2>&1 | tail -4 ) | tee $O/pytest_gate64.log
one() { env $1 timeout 300 python $R/bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 0 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('$2  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz  [%s]' % (d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['library_build_info']))"; }
for w in 64 32 128 256; do
  echo "# slice gate $w rows against the shipped library"
  one "BSGS_LIB_PATH=$B/libbsgs_hip.so" A; one "BSGS_LIB_PATH=$B/exp_gate$w/libbsgs_hip.so" B; one "BSGS_LIB_PATH=$B/exp_gate$w/libbsgs_hip.so" B; one "BSGS_LIB_PATH=$B/libbsgs_hip.so" A
done 2>&1 | tee $O/slice_gate_abba.log
cd /tmp
for w in 0 64; do
  lib=$B/libbsgs_hip.so; [ $w != 0 ] && lib=$B/exp_gate$w/libbsgs_hip.so
  rm -rf /tmp/fg; BSGS_LIB_PATH=$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fg -- python $R/bench.py --pmc-child --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv,glob,re
f=glob.glob("/tmp/fg/**/*counter_collection.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Dispatch_Id"]))
v=[float(r["Counter_Value"]) for r in rows if re.search(r"giant_pair2_kernel<\d, false, (true|false)>", r["Kernel_Name"]) and r["Counter_Name"]=="FETCH_SIZE"][-3:]
print("gate $w: raw FETCH_SIZE %.2f B per giant step" % (sum(v)/len(v)*1024/(192<<25)))
PY
done 2>&1 | tee $O/slice_gate_fetch.log
