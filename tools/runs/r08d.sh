#!/bin/bash
# round 5, GPU call 8d: the overflow fingerprint widened from 16 to 31 bits (header = 0x80000000 | bits): the tests that touch lines + overflow-set tables, A B B A at -w 35
# against the 16-bit library (build/exp_fp16), then fuller lines with the new library (2^35 points in 2.667 * 2^30 lines: load 12, 171 GiB) and the 128-byte-line table
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08d; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -x -q -k "fingerprint or overflow or direct_line or planted or census or extended_table or three_startup or any_number_of_buckets or false_positives or fuzz" 2>&1 | tail -15 ) | tee $O/pytest_fingerprint.log
OLD="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_fp16/libbsgs_hip.so"; NEW="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/libbsgs_hip.so"; B4="BSGS_LIB_PATH=$R/bsgs-cuda_amd/build/exp_before_fp/libbsgs_hip.so"
( SUSTAIN=5 STEPS=20 bash tools/abba.sh "$OLD" "$NEW" --w 35 --htsz 3221225472 --no-refquirks-leg ) 2>&1 | tee $O/abba_w35_fingerprint16_vs_31.log
run() { env $1 python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 --steps 20 --warmup 3 "${@:3}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['alu']['power'] or {}
print('$2  %.2f Gsteps/s  %.3f ms/launch  sclk %.0f MHz  over-full %s  table GiB %.1f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], p.get('sclk_MHz_mean',0), d['config']['overflow_buckets'], 0))"; }
( run "$NEW" "load12_lines64_new" --w 35 --htsz 2863311531 --layout 4
  run "$B4" "load12_lines64_before" --w 35 --htsz 2863311531 --layout 4
  run "$NEW" "load12_lines64_new" --w 35 --htsz 2863311531 --layout 4
  run "$NEW" "lines128_1.5x2e30_new" --w 35 --htsz 1610612736 --layout 5
  run "$B4" "lines128_1.5x2e30_before" --w 35 --htsz 1610612736 --layout 5
  run "$NEW" "lines128_1.5x2e30_new" --w 35 --htsz 1610612736 --layout 5 ) 2>&1 | tee $O/w35_other_shapes.log
