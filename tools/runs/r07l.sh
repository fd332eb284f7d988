#!/bin/bash
# round 5, GPU call 7l: 64-byte lines with ANY number of buckets (tile kernels <4, ., .>): parity tests at small sizes, then -w 35 on 3 * 2^30 lines of 64 bytes (192 GiB + a 16 GiB
# overflow set) against the shipped -w 35 table (1.5 * 2^30 lines of 128 bytes) on the same box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07l; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round5.py -m gpu -x -q -k "census_and_sampled or three_startup or any_number_of_buckets or twin" 2>&1 | tail -30 ) | tee $O/pytest_lines64_any.log
for CFG in "3221225472 0" "1610612736 0" "2952790016 4"; do set -- $CFG
  BSGS_BUILD_VERBOSE=1 python bench.py --w 35 --htsz $1 --layout $2 --no-cpu-baseline --no-pmc --no-solve --no-refquirks-leg --sustain-s 5 > $O/bench_w35_buckets$1.json 2> $O/bench_w35_buckets$1.err
  python - $O/bench_w35_buckets$1.json <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "%.2f G" % (d["value"] / 1e9), "sustained %.2f G" % (d["value_sustained"] / 1e9), "keys/s %.3e" % d["effective_keys_per_s"], "tiles per launch", d["roofline"]["tiles_per_launch"],
          d["config"]["table_layout"], "over-full", d["config"]["overflow_buckets"], "build %.2f s" % d["table_build"]["seconds"], "hits", d.get("false_positive_hits"), "scratch", d.get("chain_scratch", {}).get("from_reserved_group"), d["roofline"]["kernel"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
  tail -3 $O/bench_w35_buckets$1.err
done 2>&1 | tee $O/w35_lines64_any.log
