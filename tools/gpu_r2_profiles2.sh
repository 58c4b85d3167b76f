#!/bin/bash
# BN254 2^24 without the PLONK leg (per-launch averages of one size only) and the PLONK leg on its own
TAG=${TAG:-r02}
OUT=gpurun_out/prof_r2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --plonk-log-n 0"
run() { name=$1; shift; timeout 900 rocprofv3 "$@" > $OUT/$name.log 2>&1 || echo "rocprofv3 $name failed"; }
run bn254_stats --kernel-trace --stats -d $OUT/bn254_stats -o k -- $BENCH
python tools/prof_summary.py $OUT/bn254_stats/k_results.db > $OUT/${TAG}_bench24_bn254_kernel_stats.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  run bn254_$ctr --pmc $ctr --kernel-trace -d $OUT/bn254_$ctr -o p -- $BENCH
  python tools/prof_summary.py --pmc $OUT/bn254_$ctr/p_results.db > $OUT/${TAG}_bench24_bn254_pmc_${ctr}.txt
  head -6 $OUT/${TAG}_bench24_bn254_pmc_${ctr}.txt | cut -c1-180
  run plonk_$ctr --pmc $ctr --kernel-trace -d $OUT/plonk_$ctr -o p -- python tools/bench_plonk_kernels.py
  python tools/prof_summary.py --pmc $OUT/plonk_$ctr/p_results.db > $OUT/${TAG}_plonk_2p22_pmc_${ctr}.txt
  head -8 $OUT/${TAG}_plonk_2p22_pmc_${ctr}.txt | cut -c1-180
done
run plonk_stats --kernel-trace --stats -d $OUT/plonk_stats -o k -- python tools/bench_plonk_kernels.py
python tools/prof_summary.py $OUT/plonk_stats/k_results.db > $OUT/${TAG}_plonk_2p22_kernel_stats.txt
head -14 $OUT/${TAG}_plonk_2p22_kernel_stats.txt | cut -c1-180
python tools/bench_plonk_kernels.py > $OUT/${TAG}_bench_plonk_2p22.json 2>/dev/null
rm -rf $OUT/*_stats $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE
