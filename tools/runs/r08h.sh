#!/bin/bash
# round 5, GPU call 8h: Tune's new choice for large ranges -- 36 * 2^30 baby points on 3 * 2^30 lines of 64 bytes (r08g: 2.93e21 keys/s against 2.65e21 at 2^35) --
# pinned at full size (crafted centres, census, 2e5 sampled keys) and BASELINE config 3 (80-bit range, key half-way) at -w auto
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08h; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -s -k "extended_table_w35" 2>&1 | grep -v "^\[build\]" | tail -12 ) | tee $O/pytest_w35_and_36g.log
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
