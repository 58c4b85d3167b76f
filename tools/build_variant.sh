#!/bin/bash
# Build an A/B variant of the library with extra compile-time knobs, next to the shipped one:
#   tools/build_variant.sh <name> -D<KNOB>=<value>
#   tools/build_variant.sh --only "ntt_bn254 ntt_bls12381 ntt_domain plonk_bn254 plonk_bls12381" <name> -D<KNOB>=<value>
# The shipped sources carry NO compile-time knobs any more: the ones measured in rounds 3-4 (GA_ACC_LDS_PAD, GA_ACC_REGS,
# GA_ACC_THREADS_WIDE, GA_NTT_RADIX8, GA_NTT_KUP, GA_BIGFIELD_CALLS, GA_NTT_TW_UNPACKED) live in tools/exp/r04_pruned_knobs.patch --
# apply that patch (or add your own #ifdef) first; a -D that no #if in gnark_amd/csrc tests is refused below, because such a
# variant would silently equal the default build.
# -> gnark_amd/variants/libgnark_amd_<name>.so (git-ignored; it travels to the GPU box with gpurun), used with
#   GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_pad3.so python bench.py --no-cpu-baseline --no-check
# --only: the knob touches just these translation units; every other object is taken from the shipped build (gnark_amd/csrc/build,
# which must be up to date) instead of being recompiled.
set -e
only=""
if [ "$1" = "--only" ]; then only="$2"; shift 2; fi
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
for d in "$@"; do
  case "$d" in
    -D*) knob=${d#-D}; knob=${knob%%=*}
         if ! grep -rqE "^[[:space:]]*#[[:space:]]*(if|ifdef|ifndef|elif).*\b$knob\b" "$root/gnark_amd/csrc" --include='*.h' --include='*.hip'; then
           echo "build_variant.sh: no #if in gnark_amd/csrc tests $knob -- the variant would equal the default build (tools/exp/r04_pruned_knobs.patch?)" >&2; exit 2
         fi ;;
  esac
done
mkdir -p "$root/gnark_amd/variants"
bdir="$root/gnark_amd/csrc/build_$name"
if [ -n "$only" ]; then
  make -C "$root/gnark_amd/csrc" -j8 >/dev/null
  mkdir -p "$bdir"
  cp -p "$root"/gnark_amd/csrc/build/*.o "$root"/gnark_amd/csrc/build/*.d "$bdir"/
  sed -i "s#^build/#build_$name/#" "$bdir"/*.d
  touch "$bdir"/*.o          # (an object is older than its freshly edited .d file otherwise, and make would rebuild it)
  for u in $only; do rm -f "$bdir/$u.o"; done
fi
make -C "$root/gnark_amd/csrc" -j8 BUILD=build_$name TARGET=../variants/libgnark_amd_$name.so EXTRA="$*"
echo "$root/gnark_amd/variants/libgnark_amd_$name.so"
