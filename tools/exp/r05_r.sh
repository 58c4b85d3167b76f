#!/bin/bash
# round 5, GPU call R: evidence for configs 4 and 5 on the final code (kernel stats + SQ accounting of the BLS12-381 kernels, the PLONK leg under rocprofv3), the small end
# of the size sweep, the one-shot (un-pinned key: the Go shim's default) proof
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_r}; export TMPDIR=/tmp
TAG=$TAG bash tools/gpu.sh "sq:bls:bls"
d=$OUT/stats_plonk_$$; timeout 600 rocprofv3 --kernel-trace --stats -d $d -o k -- python tools/bench_plonk_kernels.py > $OUT/${TAG}_plonk_stats.log 2>&1
python tools/prof_summary.py $d/k_results.db > $OUT/${TAG}_plonk22_kernel_stats.txt 2>/dev/null; rm -rf $d; head -16 $OUT/${TAG}_plonk22_kernel_stats.txt | cut -c1-170
timeout 900 python tools/size_sweep.py --curve bn254 --logs 10,12,14,16,18,20,22 > $OUT/${TAG}_size_sweep_bn254_small.jsonl 2> $OUT/${TAG}_sweep.err; cut -c1-200 $OUT/${TAG}_size_sweep_bn254_small.jsonl
timeout 900 python tools/size_sweep.py --curve bn254 --logs 24 --one-shot --precompute -1 > $OUT/${TAG}_one_shot_bn254_2p24.jsonl 2>> $OUT/${TAG}_sweep.err; cat $OUT/${TAG}_one_shot_bn254_2p24.jsonl
