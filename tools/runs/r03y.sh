#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
( timeout 1200 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "whole_tile" 2>&1 | tail -25 ) > $O/pytest.log; cat $O/pytest.log
