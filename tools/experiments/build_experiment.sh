#!/bin/bash
# tools/experiments/build_experiment.sh <name> "<-D switches>"  ->  bsgs-cuda_amd/build/exp_<name>/libbsgs_hip.so
#
# The shipped tile kernel (csrc/giant_kernel.hip.h) carries no timing experiments.  This script copies csrc/ + host/ + the Makefile to
# build/exp_<name>/src, applies tile_kernel_experiments.patch (the *_CEILING switches -- libraries that return WRONG results and keep the
# timing --, BSGS_FULL_X, BSGS_INV_PER_WAVE and the slice gate) to the COPY, and builds the library there.  bsgs_build_info() of such a
# library names its switches ("WRONG-RESULTS:<switch>" for the ceilings); the hosts refuse to search with it.  Examples:
#   tools/experiments/build_experiment.sh nochain  "-DBSGS_EXPERIMENT -DBSGS_NOCHAIN_CEILING"      (tools/fetch_breakdown.py)
#   tools/experiments/build_experiment.sh g2cached "-DBSGS_EXPERIMENT -DBSGS_G2_CACHED_CEILING"    (tools/fetch_breakdown.py)
#   tools/experiments/build_experiment.sh gate64   "-DBSGS_SLICE_GATE=64"                           (exact results)
# then  tools/abba.sh "BSGS_LIB_PATH=bsgs-cuda_amd/build/libbsgs_hip.so" "BSGS_LIB_PATH=bsgs-cuda_amd/build/exp_<name>/libbsgs_hip.so"
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
NAME=$1; EXTRA=$2
[ -n "$NAME" ] || { echo "usage: $0 <name> \"<-D switches>\"" >&2; exit 2; }
OUT=$ROOT/bsgs-cuda_amd/build/exp_$NAME
rm -rf "$OUT/src"; mkdir -p "$OUT/src/bsgs-cuda_amd" "$OUT/src/include"
cp -r "$ROOT/bsgs-cuda_amd/csrc" "$ROOT/bsgs-cuda_amd/host" "$ROOT/bsgs-cuda_amd/Makefile" "$OUT/src/bsgs-cuda_amd/"
cp "$ROOT"/include/*.h "$OUT/src/include/"
rm -f "$OUT/src/bsgs-cuda_amd/csrc/microbench"
patch -s -p1 -d "$OUT/src/bsgs-cuda_amd" < "$HERE/tile_kernel_experiments.patch"
make -s -C "$OUT/src/bsgs-cuda_amd" -j8 ARCH=gfx950 EXTRA="$EXTRA" build/libbsgs_hip.so
cp "$OUT/src/bsgs-cuda_amd/build/libbsgs_hip.so" "$OUT/libbsgs_hip.so"
echo "$OUT/libbsgs_hip.so"
