#!/bin/bash
# Round-3 batch R: circuits that are sparse on the B side (pk.InfinityB set for most wires, as in real circuits): proof time at 2^24
# and 2^22 with 30 % / 5 % of the wires in B, checked by known dlogs at 2^22 (and 2^24 for 30 %)
OUT=gpurun_out/r3r
mkdir -p $OUT
export TMPDIR=/tmp
for D in 0.3 0.05; do
  timeout 600 python tools/size_sweep.py --curve bn254 --logs 22 --check-max 22 --proofs 5 --b-density $D >> $OUT/sweep_sparse_b.jsonl 2>> $OUT/sweep_sparse_b.err
done
timeout 900 python tools/size_sweep.py --curve bn254 --logs 24 --check-max 24 --proofs 5 --b-density 0.3 >> $OUT/sweep_sparse_b.jsonl 2>> $OUT/sweep_sparse_b.err
timeout 600 python tools/size_sweep.py --curve bn254 --logs 24 --proofs 5 --b-density 0.05 >> $OUT/sweep_sparse_b.jsonl 2>> $OUT/sweep_sparse_b.err
cat $OUT/sweep_sparse_b.jsonl; tail -5 $OUT/sweep_sparse_b.err
