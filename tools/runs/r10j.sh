#!/bin/bash
# round 6, GPU call j: the faulty builder (variant `drop`) against reference-format tables now that the builder checks that every point left its position behind; the
# builder's own tests on the shipped library
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r10j; mkdir -p $O; cd $R
export TMPDIR=/tmp
M=bsgs-cuda_amd/build_mut/drop/bsgs_mi355x
PUB=03100611c54dfef604163b8358f7b7fac13ce478e02cb224ae16d45526b25d9d4d
for flags in "-w 24 -htsz 22" "-w 30 -htsz 28"; do
  d=$(mktemp -d /tmp/mutXXXX)
  echo "== faulty builder, $flags" >> $O/faulty_builder_file_tables.log
  $M -dir $d -t 256 -b 64 -p 64 $flags -pb $PUB -pk 8000000000000000 -pke ffffffffffffffff > $d/out.txt 2> $d/err.txt; echo "exit code $?" >> $O/faulty_builder_file_tables.log
  grep -hE "verification|KEY\[|error|table build" $d/out.txt $d/err.txt | cut -c1-400 >> $O/faulty_builder_file_tables.log
done
cat $O/faulty_builder_file_tables.log
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_fullsize.py -q 2>&1 | tail -4 ) | tee $O/pytest_builder_subset.log
python bench.py --no-cpu-baseline --no-solve --no-pmc --no-refquirks-leg --steps 5 --warmup 2 --sustain-s 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table build at -w 30:', d['table_build'])" | tee $O/table_build_w30.log
