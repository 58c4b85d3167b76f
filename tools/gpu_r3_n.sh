#!/bin/bash
# Round-3 batch N: (1) small / mid-size proofs after the adaptive per-bit chunk + the Straus epilogue; (2) the --gpus N code path of
# bench.py on a 1-GPU box: N ranks share the GPU, collectives over gloo (GA_BENCH_BACKEND) -- control flow, sharded key generation,
# proof bytes; the numbers mean nothing
OUT=gpurun_out/r3n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/size_sweep.py --curve bn254 --logs 14,16,18,20,22 --proofs 20 > $OUT/sweep_small.jsonl 2> $OUT/sweep_small.err; cat $OUT/sweep_small.jsonl
timeout 300 python tools/small_profile.py --logs 16,20 > $OUT/small.jsonl 2> $OUT/small.err; cut -c1-700 $OUT/small.jsonl
timeout 200 python tools/msm_small_trace.py --log-n 20 --reps 20 > $OUT/msm20.json 2>&1; tail -1 $OUT/msm20.json | cut -c1-600
timeout 200 python tools/msm_small_trace.py --log-n 16 --reps 20 > $OUT/msm16.json 2>&1; tail -1 $OUT/msm16.json | cut -c1-600
timeout 600 python -m pytest tests -q -m gpu -x -k "msm or groth16_cubic or bsb22 or 2_10 or builder" > $OUT/gpu_subset.log 2>&1; tail -2 $OUT/gpu_subset.log
export GA_BENCH_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2ranks.json 2> $OUT/bench_2ranks.err; echo "rc=$?" >> $OUT/bench_2ranks.err
tail -1 $OUT/bench_2ranks.json | cut -c1-3000; tail -3 $OUT/bench_2ranks.err
GA_BENCH_CONFIG4=1 timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29532 bench.py --gpus 3 --steps 3 --warmup 1 > $OUT/bench_3ranks_cfg4.json 2> $OUT/bench_3ranks_cfg4.err; echo "rc=$?" >> $OUT/bench_3ranks_cfg4.err
tail -1 $OUT/bench_3ranks_cfg4.json | cut -c1-3000; tail -3 $OUT/bench_3ranks_cfg4.err
