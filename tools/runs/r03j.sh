#!/bin/bash
# round 3, GPU call J: random-line rate of the chip against the footprint (16 / 64 / 128 / 192 GiB): what a -w 34 table can get at best
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
$R/bsgs-cuda_amd/build/microbench 16384 65536 131072 196608 2>&1 | grep -E '"coop"|device' > $O/gups_vs_footprint.jsonl
cut -c1-200 $O/gups_vs_footprint.jsonl
