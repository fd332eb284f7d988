#!/bin/bash
# round 5, GPU call 7r: what a random 64-byte line costs (rate, socket power) by FOOTPRINT -- 16 MiB (the L2s), 64 ... 512 MiB (around the 256 MB memory-side cache),
# 1, 4, 16 GiB (HBM): the price list a radix-partitioned probe (keys written out by table region, each region probed while it is cache-resident) would be built on
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07r; mkdir -p $O; cd $R
( OPS="102 110 111 112 113 114 115 116 100 103" bash tools/power_ops.sh ) 2>&1 | tee $O/power_gups_by_footprint.jsonl
