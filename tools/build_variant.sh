#!/bin/bash
# Builds a diagnostic / A-B variant of the library HERE (hipcc cross-compiles, no GPU minutes) beside the product build, without touching it:
#   tools/build_variant.sh v1 "-DCAP_W8_NODE_V1"      -> capsaicin_amd/variants/libcapsaicin_hip_v1.so
# Tools and tests pick it with CAP_LIB_VARIANT=v1 (capsaicin_amd/capi.py); the .so travels to the GPU box like the product's (git-ignored).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
W=/tmp/cap_variant_$name
mkdir -p $W/capsaicin_amd $ROOT/capsaicin_amd/variants
rm -rf $W/capsaicin_amd/csrc $W/include
mkdir -p $W/capsaicin_amd/csrc
cp $ROOT/capsaicin_amd/csrc/*.hip $ROOT/capsaicin_amd/csrc/*.h $ROOT/capsaicin_amd/csrc/*.cpp $ROOT/capsaicin_amd/csrc/Makefile $W/capsaicin_amd/csrc/
cp -r $ROOT/include $W/
make -C $W/capsaicin_amd/csrc -j8 EXTRA="$*" ../libcapsaicin_hip.so > $W/build.log 2>&1 || { tail -20 $W/build.log; exit 1; }
cp $W/capsaicin_amd/libcapsaicin_hip.so $ROOT/capsaicin_amd/variants/libcapsaicin_hip_$name.so
echo "EXTRA=$*" > $ROOT/capsaicin_amd/variants/$name.flags
echo "built capsaicin_amd/variants/libcapsaicin_hip_$name.so ($*)"
