#!/bin/bash
# round 6, batch B: why does bench.py's one-caller proof read 140 ms where the A/B harness reads 132 on the same box?  per-proof times
export TAG=r06_b
OUT=gpurun_out
for k in 1 2; do
python bench.py --no-pmc --no-selftest --no-bls --plonk-log-n 0 --no-cpu-baseline --no-check --groth16-proofs 12 --detail-file $OUT/r06_b_detail_$k.json > $OUT/r06_b_bench_$k.json 2> $OUT/r06_b_bench_$k.err
python -c "
import json; d=json.load(open('$OUT/r06_b_detail_$k.json')); g=d['groth16']; print(g['ms_per_proof'], g['ms_each'], g['pipelined']['ms_per_proof'], g['one_shot_unpinned'])"
done
