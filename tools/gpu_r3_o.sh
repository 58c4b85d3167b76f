#!/bin/bash
# Round-3 batch O: small proofs after the dense-set lazy window reduction; GPU subset of the suite
OUT=gpurun_out/r3o
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/size_sweep.py --curve bn254 --logs 10,12,14,16,18,20 --proofs 20 > $OUT/sweep_small.jsonl 2> $OUT/sweep_small.err; cat $OUT/sweep_small.jsonl
timeout 300 python tools/size_sweep.py --curve bls12-381 --logs 12,14,16,20 --proofs 20 > $OUT/sweep_small_bls.jsonl 2> $OUT/sweep_small_bls.err; cat $OUT/sweep_small_bls.jsonl
timeout 300 python tools/small_profile.py --logs 12,14 > $OUT/small.jsonl 2> $OUT/small.err; cut -c1-700 $OUT/small.jsonl
timeout 900 python -m pytest tests -q -m gpu -x -k "not 2_24 and not 2_26 and not 2_22" > $OUT/gpu_subset.log 2>&1; tail -2 $OUT/gpu_subset.log
