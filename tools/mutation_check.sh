#!/bin/bash
# Builds deliberately WRONG variants of libbsgs_hip.so (never shipped: bsgs-cuda_amd/build_mut/ is git-ignored) to show that the any-bucket parity tests of
# tests/test_gpu_round6.py bite.  Each variant is a sed on a scratch copy of csrc/:
#   fp2     the probe tests the second fingerprint bit of a DIFFERENT hash bit range -> false negatives for some set-only hashes (kernels <4> and <3>)
#   carry   bucket_mul48 without its (xhi & 0xFFFF) * M >> 16 term -> a wrong bucket for M / 2^33 of all keys
# and to show that the HOST's verification of what it builds (host_engines.cpp verify_tables) catches a faulty BUILDER, not only an injected bit flip:
#   drop    the table builder skips one point in 2^20 (k - 1 = 12345 mod 2^20)      -> extended tables: the census counts fewer than w entries; reference-format tables: the builder's
#           own check that every point left its position behind (positions_written_kernel) refuses the build
# Run on the GPU:  BSGS_LIB_PATH=bsgs-cuda_amd/build_mut/<variant>/libbsgs_hip.so python -m pytest tests/test_gpu_round6.py   (expected: failures)
#                  bsgs-cuda_amd/build_mut/drop/bsgs_mi355x -w 24 -htsz 21 -ext ...   (expected: "table verification FAILED: ... census")
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in ${MUTANTS:-fp2 carry drop}; do
    W=$(mktemp -d /tmp/mut_${v}_XXXX)
    mkdir -p $W/repo && cp -r $ROOT/bsgs-cuda_amd $W/repo/ && cp -r $ROOT/include $W/repo/
    rm -rf $W/repo/bsgs-cuda_amd/build
    K=$W/repo/bsgs-cuda_amd/csrc/giant_kernel.hip.h
    case $v in
      fp2)   sed -i 's/(BK ? hdr >> ovf_fingerprint_index2(xhi) : 1u)/(BK ? hdr >> ovf_fingerprint_index2(xhi >> 1) : 1u)/' $K ;;
      carry) sed -i 's/return (u32)(((u64)xlo \* M + (((u64)(xhi \& 0xFFFFu) \* M) >> 16)) >> 32); }/return (u32)(((u64)xlo * M) >> 32); }/' $K ;;
      drop)  K=$W/repo/bsgs-cuda_amd/csrc/baby_builder.hip
             sed -i 's/        if (idx >= count) return;/        if (idx >= count || (idx \& 0xFFFFFu) == 12345u) return;/' $K ;;
    esac
    if cmp -s $K $ROOT/bsgs-cuda_amd/csrc/$(basename $K); then echo "mutation $v did not apply"; exit 1; fi
    make -s -j8 -C $W/repo/bsgs-cuda_amd build/libbsgs_hip.so build/bsgs_mi355x
    mkdir -p $ROOT/bsgs-cuda_amd/build_mut/$v
    cp $W/repo/bsgs-cuda_amd/build/libbsgs_hip.so $W/repo/bsgs-cuda_amd/build/bsgs_mi355x $ROOT/bsgs-cuda_amd/build_mut/$v/
    rm -rf $W
    echo "built variant $v"
done
