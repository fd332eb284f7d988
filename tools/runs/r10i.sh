#!/bin/bash
# round 6, GPU call i: (1) a FAULTY BUILDER (tools/mutation_check.sh, variant `drop`: one point in 2^20 is not filed) against the host's own verification, extended and file tables;
# (2) the any-bucket fuzz under three more seeds (3 x 4000 cases per line size) and the round-3 fuzz of the five image layouts on the final library (4000 cases)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10i; mkdir -p $O; cd $R
export TMPDIR=/tmp
M=bsgs-cuda_amd/build_mut/drop/bsgs_mi355x
PUB=03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d
for flags in "-w 26 -htsz 23 -ext" "-w 26 -buckets 6291456" "-w 24 -htsz 22"; do
  d=$(mktemp -d /tmp/mutXXXX)
  echo "== faulty builder, $flags" >> $O/faulty_builder_vs_host_verification.log
  $M -dir $d -t 256 -b 64 -p 64 $flags -pb $PUB -pk 8000000000000000 -pke ffffffffffffffff > $d/out.txt 2> $d/err.txt; echo "exit code $?" >> $O/faulty_builder_vs_host_verification.log
  grep -hE "verification|KEY\[" $d/out.txt $d/err.txt | cut -c1-400 >> $O/faulty_builder_vs_host_verification.log
  echo "== shipped builder, $flags" >> $O/faulty_builder_vs_host_verification.log
  d=$(mktemp -d /tmp/okXXXX)
  ./bsgs-cuda_amd/build/bsgs_mi355x -dir $d -t 256 -b 64 -p 64 $flags -pb $PUB -pk 8000000000000000 -pke 8000000fffffffff > $d/out.txt 2> $d/err.txt; echo "exit code $?" >> $O/faulty_builder_vs_host_verification.log
  grep -hE "Table verification|FAILED" $d/out.txt $d/err.txt | cut -c1-300 >> $O/faulty_builder_vs_host_verification.log
done
cat $O/faulty_builder_vs_host_verification.log
for seed in 11 22 33; do for f in 6 7; do
  BSGS_FUZZ_CASES=4000 BSGS_FUZZ_SEED=$seed python -m pytest "tests/test_gpu_round6.py::test_fuzz_any_bucket_tables_complete_hit_lists[$f]" -q -s 2>&1 | grep -E "family|passed|failed" >> $O/fuzz_any_bucket_more_seeds.log
done; done
BSGS_FUZZ_CASES=4000 BSGS_FUZZ_SEED=606 python -m pytest tests/test_gpu_round3.py::test_fuzz_random_geometries_layouts_and_flags -q 2>&1 | tail -2 >> $O/fuzz_image_layouts_4000.log
cat $O/fuzz_any_bucket_more_seeds.log $O/fuzz_image_layouts_4000.log
