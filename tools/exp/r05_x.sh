#!/bin/bash
# round 5, batch X: the task list of large MSMs ordered by exact length (GA_MSM_TASK_EXACT_MIN = 2^25 pairs, the new default) against the
# 7-bit quantised order of the closing-run build (the same library with the knob above every size), one box: table MSMs of 2^24
# points (tools/ab_kernels.py --parts msm, both groups) twice each, then the PLONK leg and a proof through bench.py
export TAG=r05_x
Q=GA_MSM_TASK_EXACT_MIN=1099511627776
bash tools/gpu.sh "abenv:quantised:msm:bn254:$Q" "abenv:exact:msm:bn254:-" "abenv:quantised_b:msm:bn254:$Q" "abenv:exact_b:msm:bn254:-"
for t in quantised exact; do
  e=""; [ "$t" = quantised ] && e=$Q
  echo "=== bench.py $t"
  env $e timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --groth16-proofs 3 --no-pipelined --no-bls --no-selftest --no-pmc > gpurun_out/r05_x_bench_$t.json 2> gpurun_out/r05_x_bench_$t.err
  tail -2 gpurun_out/r05_x_bench_$t.err
  python tools/bench_digest.py gpurun_out/r05_x_bench_$t.json | grep -E "headline|^  stages|groth16_bn254_ms|plonk_bn254|groth16 stages|plonk stages"
done
