#!/bin/bash
# PLONK row: parity tests for the device quotient / grand product + the config-5 bench (separate-kernel count and fused pipeline)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -k "plonk" > gpurun_out/pytest_plonk.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_plonk.log
tail -12 gpurun_out/pytest_plonk.log
timeout 900 python tools/bench_plonk_kernels.py > gpurun_out/bench_plonk.log 2>&1; echo "rc=$?" >> gpurun_out/bench_plonk.log
cut -c1-1500 gpurun_out/bench_plonk.log
