#!/bin/bash
# Builds deliberately WRONG variants of libbsgs_hip.so (never shipped: bsgs-cuda_amd/build_mut/ is git-ignored) to show that the any-bucket parity tests of
# tests/test_gpu_round6.py bite.  Each variant is a sed on a scratch copy of csrc/:
#   fp2     the probe tests the second fingerprint bit of a DIFFERENT hash bit range -> false negatives for some set-only hashes (kernels <4> and <3>)
#   carry   bucket_mul48 without its (xhi & 0xFFFF) * M >> 16 term -> a wrong bucket for M / 2^33 of all keys
# Run on the GPU:  BSGS_LIB_PATH=bsgs-cuda_amd/build_mut/<variant>/libbsgs_hip.so python -m pytest tests/test_gpu_round6.py   (expected: failures)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for v in fp2 carry; do
    W=$(mktemp -d /tmp/mut_${v}_XXXX)
    mkdir -p $W/repo && cp -r $ROOT/bsgs-cuda_amd $W/repo/ && cp -r $ROOT/include $W/repo/
    rm -rf $W/repo/bsgs-cuda_amd/build
    K=$W/repo/bsgs-cuda_amd/csrc/giant_kernel.hip.h
    case $v in
      fp2)   sed -i 's/(BK ? hdr >> ovf_fingerprint_index2(xhi) : 1u)/(BK ? hdr >> ovf_fingerprint_index2(xhi >> 1) : 1u)/' $K ;;
      carry) sed -i 's/return (u32)(((u64)xlo \* M + (((u64)(xhi \& 0xFFFFu) \* M) >> 16)) >> 32); }/return (u32)(((u64)xlo * M) >> 32); }/' $K ;;
    esac
    if cmp -s $K $ROOT/bsgs-cuda_amd/csrc/giant_kernel.hip.h; then echo "mutation $v did not apply"; exit 1; fi
    make -s -j8 -C $W/repo/bsgs-cuda_amd build/libbsgs_hip.so
    mkdir -p $ROOT/bsgs-cuda_amd/build_mut/$v
    cp $W/repo/bsgs-cuda_amd/build/libbsgs_hip.so $ROOT/bsgs-cuda_amd/build_mut/$v/
    rm -rf $W
    echo "built variant $v"
done
