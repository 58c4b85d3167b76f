#!/bin/bash
# round 5, batch T: window width 24 for table MSMs of 2^24 points and more (11 windows, 2^23 buckets) against the planned 22 (12 windows, 2^21 buckets),
# same box, the variant library whose plan admits c = 24 (util.hip.h: c <= 24); GA_TABLE_C=22 forces the old width
V=/root/repo/gnark_amd/variants/libgnark_amd_c24.so
export TAG=r05_t
bash tools/gpu.sh "abenv:c22:msm:bn254:GA_LIB_PATH=$V,GA_TABLE_C=22" "abenv:c24:msm:bn254:GA_LIB_PATH=$V" "abenv:c22b:msm:bn254:GA_LIB_PATH=$V,GA_TABLE_C=22" "abenv:c24b:msm:bn254:GA_LIB_PATH=$V"
for c in 22 0; do
  echo "=== bench.py GA_TABLE_C=$c"
  GA_LIB_PATH=$V GA_TABLE_C=$c timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --groth16-proofs 3 --plonk-log-n 0 --no-bls --no-selftest --no-pmc --detail-file gpurun_out/r05_t_bench_c${c}_detail.json > gpurun_out/r05_t_bench_c$c.json 2> gpurun_out/r05_t_bench_c$c.err
  tail -2 gpurun_out/r05_t_bench_c$c.err
  python tools/bench_digest.py gpurun_out/r05_t_bench_c$c.json | head -24
done
