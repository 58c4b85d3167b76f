#!/bin/bash
# round 5, GPU call G: per-vector window tables at 2^26 (a context per size), the Groth16 / MSM GPU tests on the final kernels
OUT=gpurun_out; mkdir -p $OUT; TAG=${TAG:-r05_g}
TAG=$TAG bash tools/gpu.sh "tests:groth16 or msm or exception"
timeout 1500 python tools/size_sweep.py --curve bn254 --logs 25,26 --check-max 26 > $OUT/${TAG}_size_sweep_bn254_top.jsonl 2> $OUT/${TAG}_sweep.err
tail -3 $OUT/${TAG}_sweep.err; cat $OUT/${TAG}_size_sweep_bn254_top.jsonl
timeout 300 python tools/msm_small_trace.py --log-n 20 --reps 40 --mode raw > $OUT/${TAG}_msm_2p20_raw.json 2>> $OUT/${TAG}_small.err; cat $OUT/${TAG}_msm_2p20_raw.json
