#!/bin/bash
# round 4, GPU call 6j: bucket lines of a table above 40 GiB composed of graded 4 GiB chunks (hipMemCreate / hipMemMap) instead of allocate-all / free / wait-for-the-wipe / allocate:
# the tests that use such tables, start-up times, and the tile kernel's rate on both kinds of memory (ABBA)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_round3.py tests/test_gpu_host.py -m gpu -x -q -k "w34 or 40GiB or config3 or config5 or parked or two_engines" 2>&1 | tail -8 ) | tee $O/pytest.log
for m in 1 0; do echo "# BSGS_CHUNK_LINES=$m"; BSGS_CHUNK_LINES=$m REPS=1 BSGS_BUILD_VERBOSE=1 python tools/build_bench.py 34 31 2>&1 | grep -v amdgpu.ids; done 2>&1 | tee $O/build_w34_chunks_vs_walk.log
{ echo "# -w 34 -htsz 31: A = lines by hipMalloc after the walk (BSGS_CHUNK_LINES=0), B = lines composed of graded chunks (default)"; STEPS=20 bash tools/abba.sh "BSGS_CHUNK_LINES=0" "BSGS_CHUNK_LINES=1" --w 34 --htsz 31; } 2>&1 | tee $O/abba_w34_chunk_lines.log
