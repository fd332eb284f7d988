#!/bin/bash
# round 3, GPU call L: -w 34: physically contiguous bucket lines (large page-table fragments) against ordinary pages
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
export TMPDIR=/tmp
STEPS=20 bash tools/abba.sh "BSGS_CONTIGUOUS=0" "BSGS_CONTIGUOUS=1" --w 34 --htsz 31 > $O/abba_w34_contiguous.log 2>&1
cat $O/abba_w34_contiguous.log
BSGS_CONTIGUOUS=1 python bench.py --w 34 --htsz 31 --no-cpu-baseline --no-pmc --no-solve --sustain-s 0 --steps 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['big_buffers_GiB'], d['chain_scratch'])"
