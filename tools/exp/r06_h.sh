#!/bin/bash
# round 6, batch H: ga_g16_prove_oneshot on the device -- parity tests, then the one-shot figures of bench.py's Groth16 leg (BN254, 2^24)
export TAG=r06_h
OUT=gpurun_out
tools/gpu.sh "tests:prove_oneshot or batched_witness or second_caller or abi"
python bench.py --no-pmc --no-selftest --no-bls --plonk-log-n 0 --no-cpu-baseline --no-pipelined --groth16-proofs 5 --detail-file $OUT/r06_h_detail.json > $OUT/r06_h_bench.json 2> $OUT/r06_h_bench.err
tail -3 $OUT/r06_h_bench.err
python -c "
import json; d=json.load(open('$OUT/r06_h_detail.json')); g=d['groth16']; print(g['ms_per_proof'], g['ms_each'], g.get('matches_dlog')); print(json.dumps(g['one_shot_unpinned']))"
