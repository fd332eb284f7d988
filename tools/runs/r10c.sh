#!/bin/bash
# round 6, GPU call c: after the library / host split, the test-hook separation and the host's table verification: smoke, the whole GPU suite, the default bench, and
# what the verification costs at -w 30 (files), -w 34 -htsz 31 and 36 * 2^30 points (-w auto), the key near the start of the range
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10c; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.log
( timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) | tee $O/pytest_gpu.log
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4 | tee $O/bench_default.time
python tools/config3_run.py 0.0002 /tmp/cfg3a "-w 34 -htsz 31" > $O/verify_cost_w34.json 2>&1
python tools/config3_run.py 0.0002 /tmp/cfg3b "-w auto" > $O/verify_cost_w_auto.json 2>&1
mkdir -p /tmp/w30 && ( ./bsgs-cuda_amd/build/bsgs_mi355x -dir /tmp/w30 -t 256 -b 256 -p 256 -w 30 -htsz 28 -pb 03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d -pk 8000000000000000 -pke ffffffffffffffff 2>&1 | grep -E "startup|verification|KEY|Job time" ) > $O/verify_cost_w30.log
tail -3 $O/verify_cost_w30.log; python - <<PY
import json
for f in ("verify_cost_w34", "verify_cost_w_auto"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["found"], d.get("job_time_s"), d["verification"], [l for l in d["startup"] if "verification" in l])
    except Exception as e:
        print(f, "unreadable", e)
PY
