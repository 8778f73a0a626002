#!/bin/bash
# An experiment build of the library next to the product one: scripts/build_variant.sh NAME "EXTRA hipcc flags" [objects to rebuild ...]
#   -> build/libxfmr_hip_NAME.so (load it with XFMR_HIP_LIB=build/libxfmr_hip_NAME.so; scripts/ab_bench.sh alternates two builds
#   on one box). Only the named objects (default: all) are recompiled with the extra flags; the rest are the tree's objects.
set -e
name=$1; extra=$2; shift 2 || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
d=/tmp/xfbuild_$name
rm -rf $d && mkdir -p $d/transformer-recommenders_amd/csrc $d/transformer-recommenders_amd/xfmr_rec_amd $d/include $d/scripts
cp $ROOT/transformer-recommenders_amd/csrc/*.{hip,h,inc,o} $ROOT/transformer-recommenders_amd/csrc/Makefile $d/transformer-recommenders_amd/csrc/
cp $ROOT/include/*.h $d/include/ && cp $ROOT/scripts/check_isa.py $d/scripts/
cd $d/transformer-recommenders_amd/csrc
if [ $# -gt 0 ]; then rm -f "$@"; else rm -f *.o; fi
make -j8 EXTRA="$extra" ISA_AUDIT=${ISA_AUDIT:-1} > $d/make.log 2>&1 || { tail -20 $d/make.log; exit 1; }
mkdir -p $ROOT/build && cp $d/transformer-recommenders_amd/xfmr_rec_amd/libxfmr_hip.so $ROOT/build/libxfmr_hip_$name.so
echo "built build/libxfmr_hip_$name.so"
