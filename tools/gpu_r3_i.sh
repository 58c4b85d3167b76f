#!/bin/bash
# Round-3 batch I: Groth16 proof time against the circuit size (2^16 .. 2^26 constraints), sizes above the headline checked by known dlogs
OUT=gpurun_out/r3i
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python tools/size_sweep.py --curve bn254 --logs 16,18,20,22,24 > $OUT/sweep_bn254.jsonl 2> $OUT/sweep_bn254.err; echo "rc=$?" >> $OUT/sweep_bn254.err
timeout 900 python tools/size_sweep.py --curve bn254 --logs 25,26 --check-max 26 > $OUT/sweep_bn254_big.jsonl 2> $OUT/sweep_bn254_big.err; echo "rc=$?" >> $OUT/sweep_bn254_big.err
timeout 600 python tools/size_sweep.py --curve bls12-381 --logs 16,20,22,24,25 > $OUT/sweep_bls.jsonl 2> $OUT/sweep_bls.err; echo "rc=$?" >> $OUT/sweep_bls.err
cat $OUT/sweep_*.jsonl; tail -3 $OUT/sweep_*.err
