#!/bin/bash
# round 4, GPU call 6h: the new round-4 test (reference default geometry), and bench.py end to end with every new field (verification, refquirks leg, table_build,
# settled counter children + corrected traffic, measured + cold solve)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -15 ) | tee $O/pytest.log
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("$O/bench_w30.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("value %.2f G sustained %.2f G ms/step %.2f" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, d["ms_per_step"]))
print("verification", d["table_checksum_equal"], d["replica_hits_equal"], d["verification"]["verification_launch"])
print("refquirks", json.dumps(d["refquirks"])[:400])
print("table_build", d["table_build"])
print("traffic", r["traffic"], r.get("traffic_over_algorithmic"), r.get("fetch_breakdown_B_per_step"))
m=r["traffic_measured_this_run"]; print("kernel_trace", m.get("kernel_trace")); print("calib", m.get("calibration_ratios"))
for k,v in m["passes"].items(): print(k, {a:b for a,b in v.items() if a!="calibration"})
print("solve", json.dumps(d["measured_solve"])[:1500])
PY
