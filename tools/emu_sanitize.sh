#!/bin/bash
# Sanitizer passes over the kernel logic and the host code: the unmodified product sources are compiled against the functional HIP
# emulation (tests/emu) with -fsanitize=undefined or -fsanitize=address,undefined, out of tree, and the emulation tests run on that
# build (GA_EMU_LIB_PATH, tests/conftest.py).  Device buffers are plain host allocations under the emulation, so an out-of-bounds
# access of a KERNEL is an ASAN report, too.  TEST INFRASTRUCTURE; takes ~6 min (ubsan) / ~45 min (asan) on 8 cores.
#   tools/emu_sanitize.sh ubsan|asan [pytest -k expression]
set -e
KIND=${1:-ubsan}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
HERE=$ROOT/tests/emu
SRC=$ROOT/gnark_amd/csrc
WORK=${TMPDIR:-/tmp}/ga_san
OUT=$WORK/build_$KIND
mkdir -p $OUT
if [ "$KIND" = "asan" ]; then SANF="-fsanitize=address,undefined -fno-omit-frame-pointer"; else SANF="-fsanitize=undefined -fno-sanitize-recover=undefined"; fi
FLAGS="-O1 -g -std=c++17 -fPIC -I$HERE/include -I$SRC -I$ROOT/include -w $SANF"
pids=()
for f in abi groth16 hash_to_field plonk_bn254 plonk_bls12381 ntt_domain msm_bn254_g1 msm_bn254_g2 msm_bls12381_g1 msm_bls12381_g2 ntt_bn254 ntt_bls12381 util_bn254 util_bls12381; do
  g++ $FLAGS -x c++ -c "$SRC/$f.hip" -o "$OUT/$f.o" &
  pids+=($!)
done
g++ $FLAGS -c "$HERE/emu_impl.cpp" -o "$OUT/emu_impl.o" &
pids+=($!)
for p in "${pids[@]}"; do wait $p; done
g++ -shared $SANF -o $WORK/libgnark_amd_emu_$KIND.so "$OUT"/*.o -lpthread
export GA_EMU_LIB_PATH=$WORK/libgnark_amd_emu_$KIND.so
if [ "$KIND" = "asan" ]; then
  export LD_PRELOAD=$(g++ -print-file-name=libasan.so)
  export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:log_path=$WORK/asan_log
fi
export UBSAN_OPTIONS=print_stacktrace=1
cd $ROOT
# (asan: the exception-barrier tests throw C++ exceptions inside a library that a Python process loaded next to an LD_PRELOADed
# libasan -- its __cxa_throw interceptor has no real function to forward to there and aborts; they run under ubsan and in the plain build)
SEL="$2"
if [ "$KIND" = "asan" ]; then SEL="${SEL:+($SEL) and }not exception_barrier"; fi
python -m pytest tests/test_emu_kernels.py -q -n 8 -p no:cacheprovider ${SEL:+-k "$SEL"}
