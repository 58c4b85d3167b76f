#!/bin/bash
# round 6, batch D: the 35 ms hiccup in the 2nd timed proof of bench.py's Groth16 leg when the instance carries its dlogs: the script's
# garbage collector?  (GA_BENCH_GC=1 keeps it on) or the call count? (GA_BENCH_G16_WARMUP)
export TAG=r06_d
OUT=gpurun_out
for k in gc_on gc_off gc_on_warm4; do
export GA_BENCH_GC=0 GA_BENCH_G16_WARMUP=2
[ "$k" = "gc_on" ] && export GA_BENCH_GC=1
[ "$k" = "gc_on_warm4" ] && export GA_BENCH_GC=1 GA_BENCH_G16_WARMUP=4
python bench.py --no-pmc --no-selftest --no-bls --plonk-log-n 0 --no-cpu-baseline --no-pipelined --groth16-proofs 8 --detail-file $OUT/r06_d_detail_$k.json > $OUT/r06_d_bench_$k.json 2> $OUT/r06_d_bench_$k.err
python -c "
import json; d=json.load(open('$OUT/r06_d_detail_$k.json')); g=d['groth16']; print('$k', d['ms_per_step'], g['ms_per_proof'], g['ms_each'], g.get('matches_dlog'))"
done
