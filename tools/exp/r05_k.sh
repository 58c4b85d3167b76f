#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_k}; export TMPDIR=/tmp
timeout 900 python tools/size_sweep.py --curve bls12-381 --logs 24,25 --check-max 25 > $OUT/${TAG}_size_sweep_bls12381.jsonl 2>> $OUT/${TAG}_sweep.err
cat $OUT/${TAG}_size_sweep_bls12381.jsonl
timeout 1500 python tools/size_sweep.py --curve bn254 --logs 16,18,20,22,24,25,26 --check-max 26 > $OUT/${TAG}_size_sweep_bn254.jsonl 2>> $OUT/${TAG}_sweep.err
cat $OUT/${TAG}_size_sweep_bn254.jsonl
