#!/bin/bash
# round 5, GPU call 7o: lines above 0.6 of the HBM are placed without a reserved memory group (r07n) and Tune names 3 * 2^30 lines of 64 bytes for -w 35:
# the new full-size -w 35 tests (both line sizes: crafted centres, census, sampled membership), -w 34 and the any-bucket-count tests again, then BASELINE config 3
# (80-bit range, key half-way) at Tune's choice
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07o; mkdir -p $O; cd $R
export TMPDIR=/tmp
( BSGS_BUILD_VERBOSE=1 timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round5.py -m gpu -x -q -s -k "extended_table_w35 or extended_table_w34 or any_number_of_buckets or census_and_sampled" 2>&1 | grep -v "^\[build\]" | tail -40 ) | tee $O/pytest_w35.log
( python tools/config3_run.py 0.5 /tmp/cfg3 "-w auto" ) 2>&1 | tee $O/config3_80bit_w_auto.json
