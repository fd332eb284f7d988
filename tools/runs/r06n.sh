#!/bin/bash
# round 4, GPU call 6n: the profile round of the shipped library (kernel trace + stats next to a plain process, PMC passes), summaries -> gpurun_out/prof_r06n
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export TMPDIR=/tmp
bash tools/profile_round.sh r06n 2>&1 | tail -40
python tools/pmc_traffic.py gpurun_out/prof_r06n gpurun_out/prof_r06n/pmc_traffic.json $((192 << 25)) 2>&1 | tail -5
