#!/bin/bash
# tools/power_ops.sh: socket power and shader clock while ONE instruction mix keeps all SIMDs busy (energy per operation).
R=${GRAFT_REPO_ROOT:-$PWD}
MB=$R/bsgs-cuda_amd/build/microbench
for op in ${OPS:-3 6 0 1 4 9 100 101 102}; do
  $MB power $op 6 > /tmp/po.json &
  P=$!
  sleep 3.5
  S=$(/opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' ')
  wait $P
  echo "$(cat /tmp/po.json) smi: $S"
done
