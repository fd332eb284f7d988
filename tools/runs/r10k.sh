#!/bin/bash
# round 6, GPU call k: smoke, the whole GPU suite and the driver command on the final HEAD (positions check in the reference-format builder, eight-engine host test, five CPU-baseline runs)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10k; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
( timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) | tee $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); c=d['cpu_baseline']; b=c['best_effort']
print('%.2f G  %.2f ms  frac %.3f  norm %.2f G' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['value_clock_normalised']/1e9))
print('cpu port %.1f M spread %.3f %s | fast %.1f M spread %.3f %s' % (c['value']/1e6, c['spread'], c['sample'][60:150], b['value']/1e6, b['spread'], b['sample'][:90]))"
