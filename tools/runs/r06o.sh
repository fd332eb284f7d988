#!/bin/bash
# round 4, GPU call 6o: the long configurations on the final code: config 3 for real (80-bit range, -w 34, key half-way in), config 4 (1000 keys), two ranks on one GPU
# with the verification fields, one rank under RCCL
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
export TMPDIR=/tmp
python tools/config3_run.py 0.5 /tmp/cfg3 2>&1 | tail -1 | tee $O/config3_80bit_solve.json
python tools/config4_run.py 1000 /tmp/cfg4 2>&1 | tail -1 | tee $O/config4_1000keys.json
python bench.py --gpus 2 --same-device --w 26 --htsz 25 --no-pmc --no-solve --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_two_ranks_same_device_w26.json
BSGS_DIST_FORCE=1 python bench.py --no-pmc --no-solve --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_w30_one_rank_under_rccl.json
python - <<PY
import json
for n in ("bench_two_ranks_same_device_w26","bench_w30_one_rank_under_rccl"):
    d=json.loads(open("$O/%s.json"%n).read())
    print(n, "%.2f G" % (d["value"]/1e9), d["n_gpus"], d["config"]["backend"], d["table_checksum_equal"], d["replica_hits_equal"], d["table_broadcast_GBps"], d["table_broadcast_frac_of_xgmi_link"], [ (r["rank"], round(r["giant_steps_per_s"]/1e9,2), r["table_checksums"][0]) for r in d["per_rank"]])
PY
