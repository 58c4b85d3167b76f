#!/bin/bash
# Builds the CPU oracle (test infrastructure): oracle/liboracle.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
# oracle.c: the checker (-O2).  msm_fast.c: the CPU baseline bench.py quotes (-O3; mulx/adx are in every x86 since 2014 -- no
# -march=native: the .so is built in the GPU-less container and travels to the GPU box, whose CPU is another model).
gcc -O2 -std=gnu11 -fPIC -pthread -Wall -Wno-unused-function -c "$HERE/oracle.c" -o "$HERE/oracle.o"
# ROCm's clang (same image here and on the GPU box) compiles the 128-bit product chains ~1.4x faster than gcc 11; gcc if it is missing
FASTCC=/opt/rocm/lib/llvm/bin/clang
[ -x "$FASTCC" ] || FASTCC=gcc
$FASTCC -O3 -mbmi2 -madx -std=gnu11 -fPIC -pthread -Wall -Wno-unused-function -c "$HERE/msm_fast.c" -o "$HERE/msm_fast.o"
gcc -shared -pthread "$HERE/oracle.o" "$HERE/msm_fast.o" -o "$HERE/liboracle.so"
echo "built $HERE/liboracle.so"
