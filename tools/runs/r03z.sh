#!/bin/bash
# round 3, GPU call Z: FOUR ranks on one GPU over gloo (config 5's code path at N = 4): union of the ranks' hits == one process's hits
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
export TMPDIR=/tmp
C="--w 26 --htsz 25 --tiles-per-launch 48 --warmup 1 --warmup-s 0 --sustain-s 0 --no-cpu-baseline --no-solve --no-pmc"
python bench.py $C --steps 12 --dump-hits $O/one.json > $O/bench_one.json 2> $O/one.err
python bench.py $C --steps 3 --gpus 4 --same-device --dump-hits $O/four.json > $O/bench_four.json 2> $O/four.err
python - <<PY
import json
a=json.load(open("$O/one.json")); b=json.load(open("$O/four.json"))
lo,hi=4*48,13*48
want=sorted(tuple(x) for x in a["hits"] if lo<=x[0]<hi); have=sorted(tuple(x) for x in b["hits"] if lo<=x[0]<hi)
print("ranks", b["ranks"], "launches per rank", b["per_rank_launches"], "hits", len(want), len(have), "EQUAL" if want==have else "DIFFERENT")
d=json.loads(open("$O/bench_four.json").read().strip().splitlines()[-1]); print("%.2f G aggregate, n_gpus %d, rccl_ranks %d, bcast %.2f s" % (d["value"]/1e9, d["n_gpus"], d["rccl_ranks"], d["table_broadcast_s"]))
PY
tail -2 $O/four.err
