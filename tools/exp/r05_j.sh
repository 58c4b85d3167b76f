#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_j}; export TMPDIR=/tmp
timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 40 --mode both > $OUT/${TAG}_msm_2p20.json 2>> $OUT/${TAG}_small.err; cat $OUT/${TAG}_msm_2p20.json
timeout 900 python tools/size_sweep.py --curve bls12-381 --logs 24 > $OUT/${TAG}_size_sweep_bls12381.jsonl 2>> $OUT/${TAG}_sweep.err
cat $OUT/${TAG}_size_sweep_bls12381.jsonl
timeout 900 python tools/size_sweep.py --curve bn254 --logs 16,20 > $OUT/${TAG}_size_sweep_bn254.jsonl 2>> $OUT/${TAG}_sweep.err
cat $OUT/${TAG}_size_sweep_bn254.jsonl
rocm-smi --showclocks 2>/dev/null | head -20
