#!/bin/bash
# TEST INFRASTRUCTURE: compile the *unmodified* product sources against the functional HIP emulation in
# tests/emu/include so that kernel logic can be exercised on a CPU-only box.  Output: tests/emu/libgnark_amd_emu.so
# (never loaded by the gnark_amd package).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../gnark_amd/csrc"
OUT="$HERE/build"
mkdir -p "$OUT"
# one builder at a time (pytest-xdist workers all ask for the library at session start)
exec 9>"$OUT/.lock"
flock 9
FLAGS="-O2 -g0 -std=c++17 -fPIC -I$HERE/include -I$SRC -I$HERE/../../include -w"
pids=()
for f in abi groth16 hash_to_field plonk_bn254 plonk_bls12381 ntt_domain msm_bn254_g1 msm_bn254_g2 msm_bls12381_g1 msm_bls12381_g2 ntt_bn254 ntt_bls12381 util_bn254 util_bls12381; do
  if [ ! -f "$OUT/$f.o" ] || [ -n "$(find "$SRC" "$HERE/include" -newer "$OUT/$f.o" \( -name '*.hip.h' -o -name '*.h' -o -name '*.hpp' -o -name "$f.hip" \) | head -1)" ]; then
    g++ $FLAGS -x c++ -c "$SRC/$f.hip" -o "$OUT/$f.o" &
    pids+=($!)
  fi
done
g++ $FLAGS -c "$HERE/emu_impl.cpp" -o "$OUT/emu_impl.o" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
g++ -shared -o "$HERE/libgnark_amd_emu.so.tmp" "$OUT"/*.o -lpthread
mv -f "$HERE/libgnark_amd_emu.so.tmp" "$HERE/libgnark_amd_emu.so"
echo "built $HERE/libgnark_amd_emu.so"
