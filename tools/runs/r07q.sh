#!/bin/bash
# round 5, GPU call 7q: the Comba product with its carry-outs counted on the SCALAR unit (fe_mul512_s: bit-sliced pairs / triples of carry masks, 41 vector carry steps per
# product instead of 62): exact? (salucheck: 2^18 operand pairs incl. all-ones words, half-active waves) and faster? (sustained rate, socket power and clock of the product
# alone and of product + fold, next to the shipped forms, alternating)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r07q; mkdir -p $O; cd $R
MB=$R/bsgs-cuda_amd/build/microbench
( $MB salucheck; echo "salucheck rc $?" ) 2>&1 | tee $O/salucheck.log
( OPS="201 206 200 207 207 200 206 201" bash tools/power_ops.sh ) 2>&1 | tee $O/power_ops_salu_carries.jsonl
