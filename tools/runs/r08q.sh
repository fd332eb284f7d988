#!/bin/bash
# round 5, GPU call 8q: 192 GiB of HBM from the driver -- one hipMalloc (3.9 s inside the start-up of the large tables, r08j) against pieces taken by several host threads at once
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r08q; mkdir -p $O; cd $R
( timeout 300 ./bsgs-cuda_amd/build/alloc_parallel 192 ) 2>&1 | tee $O/alloc_192GiB.jsonl
