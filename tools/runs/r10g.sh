#!/bin/bash
# round 6, GPU call g: the overflow-list tail test with its wider tail, the driver's command with 6-second CPU-baseline runs, and BASELINE configs 4 and 3 on the final host
# (split into translation units, verifies its tables, rewrites the checkpoint at every job end)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10g; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -m pytest "tests/test_gpu_round6.py::test_overflow_list_regions_spill_into_the_shared_tail" tests/test_gpu_host.py -q 2>&1 | tail -5 ) | tee $O/pytest_subset.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); c=d['cpu_baseline']; b=c['best_effort']
print('%.2f G  %.2f ms  frac %.3f  norm %.2f G' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['roofline']['value_clock_normalised']/1e9))
print('cpu port %.1f M spread %.3f %s | fast %.1f M spread %.3f %s' % (c['value']/1e6, c['spread'], c['repeats'], b['value']/1e6, b['spread'], b['repeats']))" | tee $O/bench_summary.log
python tools/config4_run.py 1000 /tmp/cfg4 > $O/config4_1000keys.json 2> $O/config4.err; tail -c 600 $O/config4_1000keys.json
python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" > $O/config3_80bit_w_auto.json 2>&1; python -c "
import json; d=json.loads(open('$O/config3_80bit_w_auto.json').read().strip().splitlines()[-1]); print('config 3: found %s job %.1f s wall %.1f s  %.2f G' % (d['found'], d['job_time_s'], d['process_wall_s_incl_table_build'], d['giant_steps_per_s']/1e9)); print(d['verification'])"
