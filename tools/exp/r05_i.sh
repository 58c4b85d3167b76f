#!/bin/bash
# round 5, GPU call I: after the scratch savings (shared sort temporaries, capped growth slack), the onesweep task sort and the new table budget
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_i}; export TMPDIR=/tmp
TAG=$TAG bash tools/gpu.sh "tests:groth16 or msm or exception or plonk"
timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 40 --mode both > $OUT/${TAG}_msm_2p20.json 2>> $OUT/${TAG}_small.err; cat $OUT/${TAG}_msm_2p20.json
timeout 300 python tools/msm_small_trace.py --log-n 16 --reps 40 --mode both > $OUT/${TAG}_msm_2p16.json 2>> $OUT/${TAG}_small.err; cat $OUT/${TAG}_msm_2p16.json
timeout 1500 python tools/size_sweep.py --curve bn254 --logs 16,20,24,25,26 --check-max 26 > $OUT/${TAG}_size_sweep_bn254.jsonl 2> $OUT/${TAG}_sweep.err
tail -3 $OUT/${TAG}_sweep.err; cat $OUT/${TAG}_size_sweep_bn254.jsonl
timeout 900 python tools/size_sweep.py --curve bls12-381 --logs 24,25 --check-max 25 > $OUT/${TAG}_size_sweep_bls12381.jsonl 2>> $OUT/${TAG}_sweep.err
cat $OUT/${TAG}_size_sweep_bls12381.jsonl
