#!/bin/bash
# round 5, GPU call 8o: on the final HEAD -- BASELINE config 4 (1000 keys, 64-bit range, -w 30 -htsz 28) and BASELINE config 3 with the flags as BASELINE.json words them
# (-w 34 -htsz 31: 2^34 points, 128 GiB of lines; round 4: 225.5 s) beside Tune's choice (r08k: 103.7 s)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08o; mkdir -p $O; cd $R
export TMPDIR=/tmp
( python tools/config4_run.py 1000 /tmp/cfg4o ) 2>&1 | tee $O/config4_1000keys.json
( python tools/config3_run.py 0.5 /tmp/cfg3o "-w 34 -htsz 31" ) 2>&1 | tee $O/config3_80bit_w34_as_worded.json
