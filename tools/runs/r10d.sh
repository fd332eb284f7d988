#!/bin/bash
# round 6, GPU call d: the whole GPU suite on the register-work HEAD (incl. the 8-rank same-device line test), the builder's stages with the rewritten ext_refine_kernel
# (36 * 2^30 points on 64-byte lines, 2^35 on 128-byte lines), and -- one box, back to back -- bench.py at -w 34 -htsz 31 against the C++ host's 80-bit search on the same table
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10d; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) | tee $O/pytest_gpu.log
for spec in "38654705664 3221225472 4" "34359738368 1610612736 5"; do
  set -- $spec
  echo "== w $1 buckets $2 layout $3" >> $O/builder_stages.log
  BSGS_BUILD_VERBOSE=1 python -c "
import torch, sys
sys.path.insert(0, 'bsgs-cuda_amd')
import pybsgs, time
d = pybsgs.Device(0)
t0 = time.time(); d.build_baby_table_ext($1, $2, $3); print('build wall %.2f s' % (time.time() - t0), file=sys.stderr)
c = d.table_census(); print('census', c, file=sys.stderr); assert c['total'] == $1 and c['malformed_lines'] == 0 and c['unsorted_lines'] == 0
d.close()" 2>&1 | grep -E "build\]|build wall|census" >> $O/builder_stages.log
done
cat $O/builder_stages.log
( python bench.py --w 34 --htsz 31 --no-cpu-baseline --no-solve --no-pmc --no-refquirks-leg > $O/bench_w34.json 2> $O/bench_w34.err ); python -c "
import json; d=json.loads(open('$O/bench_w34.json').read().strip().splitlines()[-1]); print('bench -w 34: %.3f G, %.2f ms/launch, %s tiles per launch, sclk %s' % (d['value']/1e9, d['ms_per_step'], d['config']['tiles_per_step'], d['roofline'].get('sclk_MHz_sampled')))" | tee $O/host_vs_bench.log
python tools/config3_run.py 0.15 /tmp/cfg3 "-w 34 -htsz 31" > $O/host_w34_80bit.json 2>&1
python -c "
import json; d=json.loads(open('$O/host_w34_80bit.json').read().strip().splitlines()[-1]); print('host  -w 34: %.3f G over %d tiles, job %.1f s, found %s' % (d['giant_steps_per_s']/1e9, d['tiles'], d['job_time_s'], d['found'])); print(d['verification'])" | tee -a $O/host_vs_bench.log
