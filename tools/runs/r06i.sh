#!/bin/bash
# round 4, GPU call 6i: bench.py default (does the trace child now sit at the parent's operating point?), the reference's own default geometry, config 2, -w 34
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
export TMPDIR=/tmp
python bench.py > $O/bench_w30.json 2> $O/bench_w30.err; echo "bench rc $?"
python bench.py -t 256 -b 132 -p 400 --w 25 --htsz 25 --no-pmc --no-solve --no-cpu-baseline > $O/bench_reference_defaults.json 2> $O/bench_refdef.err; echo "rc $?"
python bench.py --w 26 --htsz 25 --no-pmc --no-solve --no-cpu-baseline > $O/bench_w26_config2.json 2> $O/bench_w26.err; echo "rc $?"
python bench.py --w 34 --htsz 31 --no-pmc --no-solve --no-cpu-baseline > $O/bench_w34.json 2> $O/bench_w34.err; echo "rc $?"
python - <<PY
import json
for n in ("bench_w30","bench_reference_defaults","bench_w26_config2","bench_w34"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "FAILED", e); continue
    r=d["roofline"]
    print(n, "value %.2f G sustained %.2f G ms/launch %.2f tpl %d refq %.4f build %s setup %.1f" % (d["value"]/1e9, (d["value_sustained"] or 0)/1e9, r["avg_launch_ms"], r["tiles_per_launch"], d["refquirks"]["ratio_to_value"], d["table_build"], d["setup_s"]))
    m=r.get("traffic_measured_this_run")
    if m: print("   kernel_trace", {k:v for k,v in m.get("kernel_trace",{}).items() if k!="how"}); print("   traffic x%.3f" % r.get("traffic_over_algorithmic",0), r.get("fetch_breakdown_B_per_step"), "VALU/step", m.get("valu_instructions_per_step"), "busy", m.get("valu_busy_percent"))
    if d.get("measured_solve"): print("   solve", d["measured_solve"]["value"], "cold", d["cold_time_to_solve_s"])
PY
