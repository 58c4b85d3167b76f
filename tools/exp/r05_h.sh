#!/bin/bash
# round 5, GPU call H: 2^26 with per-vector tables (fresh contexts); the task stage of a 2^20 raw MSM kernel by kernel (MIN_SEG 256 vs 64)
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_h}; export TMPDIR=/tmp
timeout 1500 python tools/size_sweep.py --curve bn254 --logs 26 --check-max 26 > $OUT/${TAG}_size_sweep_bn254_2p26.jsonl 2> $OUT/${TAG}_sweep.err
tail -3 $OUT/${TAG}_sweep.err; cat $OUT/${TAG}_size_sweep_bn254_2p26.jsonl
for seg in 256 64; do
  d=$OUT/trace_$seg; rm -rf $d
  GA_MSM_MIN_SEG=$seg timeout 300 rocprofv3 --kernel-trace --stats -d $d -o k -- python tools/msm_small_trace.py --log-n 20 --reps 20 --mode raw > $OUT/${TAG}_trace_$seg.log 2>&1
  python tools/prof_summary.py $d/k_results.db > $OUT/${TAG}_msm_2p20_raw_kernels_seg$seg.txt 2>/dev/null; rm -rf $d
  head -30 $OUT/${TAG}_msm_2p20_raw_kernels_seg$seg.txt | cut -c1-180
done
