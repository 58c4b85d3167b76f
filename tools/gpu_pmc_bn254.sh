#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, --kernel-trace only) of the BN254 2^24 bench without the PLONK and two-caller legs
TAG=${TAG:-r02_h}
OUT=gpurun_out/prof_r2
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/bn254_$ctr -o p -- $BENCH > $OUT/bn254_$ctr.log 2>&1 || echo "rocprofv3 $ctr failed"
  python tools/prof_summary.py --pmc $OUT/bn254_$ctr/p_results.db > $OUT/${TAG}_bench24_bn254_pmc_${ctr}.txt
  grep -E "accumulate29_kernel|ntt_pass29r4|radix" $OUT/${TAG}_bench24_bn254_pmc_${ctr}.txt | head -6 | cut -c1-150
  rm -rf $OUT/bn254_$ctr
done
