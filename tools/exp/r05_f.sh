#!/bin/bash
# round 5, GPU call F: the whole -m gpu suite on the new code, then the size sweep at the top end (per-vector tables at 2^26), the 2^20 raw MSM
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_f}
TAG=$TAG bash tools/gpu.sh tests
timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 40 --mode raw > $OUT/${TAG}_msm_2p20_raw.json 2>> $OUT/${TAG}_small.err; cat $OUT/${TAG}_msm_2p20_raw.json
timeout 1500 python tools/size_sweep.py --curve bn254 --logs 24,25,26 --check-max 26 > $OUT/${TAG}_size_sweep_bn254_top.jsonl 2> $OUT/${TAG}_sweep.err
tail -3 $OUT/${TAG}_sweep.err; cat $OUT/${TAG}_size_sweep_bn254_top.jsonl
