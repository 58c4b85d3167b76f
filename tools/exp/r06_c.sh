#!/bin/bash
# round 6, batch C: one box -- bench.py's Groth16 leg with the oracle check (instance made with its dlogs) / without / with again
export TAG=r06_c
OUT=gpurun_out
for k in check nocheck check2; do
fl=""; [ "$k" = "nocheck" ] && fl="--no-check"
python bench.py --no-pmc --no-selftest --no-bls --plonk-log-n 0 --no-cpu-baseline $fl --groth16-proofs 8 --detail-file $OUT/r06_c_detail_$k.json > $OUT/r06_c_bench_$k.json 2> $OUT/r06_c_bench_$k.err
python -c "
import json; d=json.load(open('$OUT/r06_c_detail_$k.json')); g=d['groth16']; print('$k', d['ms_per_step'], g['ms_per_proof'], g['ms_each'], g['pipelined']['ms_per_proof'], g['ms_per_proof_profiled_single_lane'], g.get('matches_dlog'))"
done
