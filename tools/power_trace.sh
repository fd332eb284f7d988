#!/bin/bash
# tools/power_trace.sh [bench args]: run bench.py with a long timed region and sample rocm-smi (power, clocks) twice a second.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/power_trace.log
mkdir -p $R/gpurun_out
python $R/bench.py --no-cpu-baseline --steps 120 --warmup 5 "$@" > $R/gpurun_out/power_trace_bench.json 2>/dev/null &
BP=$!
: > $OUT
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> $OUT
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" >> $OUT
  sleep 0.5
done
wait $BP
grep -E "sclk|Power" $OUT | awk '{print}' | sort | uniq -c | sort -rn | head -30
cut -c1-200 $R/gpurun_out/power_trace_bench.json
