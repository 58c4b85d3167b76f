#!/bin/bash
# Builds the CPU oracle (test infrastructure): oracle/liboracle.so
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
gcc -O2 -std=gnu11 -shared -fPIC -pthread -Wall -Wno-unused-function "$HERE/oracle.c" -o "$HERE/liboracle.so"
echo "built $HERE/liboracle.so"
