#!/bin/bash
# Build an A/B variant of the library with extra compile-time knobs, next to the shipped one:
#   tools/build_variant.sh pad3 -DGA_ACC_LDS_PAD=3072
# -> gnark_amd/variants/libgnark_amd_pad3.so (git-ignored; it travels to the GPU box with gpurun), used with
#   GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_pad3.so python bench.py --no-cpu-baseline --no-check
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/gnark_amd/variants"
make -C "$root/gnark_amd/csrc" -j16 BUILD=build_$name TARGET=../variants/libgnark_amd_$name.so EXTRA="$*"
echo "$root/gnark_amd/variants/libgnark_amd_$name.so"
