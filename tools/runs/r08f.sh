#!/bin/bash
# round 5, GPU call 8f: BASELINE config 3 (80-bit range, key half-way) at Tune's choice again, now that the headers of over-full lines carry the overflow fingerprint
# (r07o: 117.4 s at 37.5 G before it); then the bench line at -w 35 on the same table shape
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08f; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
( python bench.py --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 10 --steps 20 --warmup 3 --w 35 --htsz 3221225472 --layout 4 2>/dev/null | tail -1 ) > $O/bench_w35_lines64.json
python -c "
import json; d=json.load(open('$O/bench_w35_lines64.json')); print('w35 bench: %.2f G giant-steps/s, %.3f ms/launch, frac %.3f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
