#!/bin/bash
# round 6, batch E: `python bench.py --gpus 2` typed as the driver types it, on a 1-GPU box: (1) backend nccl (the default): must refuse
# loudly -- one JSON error line, non-zero exit code; (2) GA_BENCH_BACKEND=gloo: bench.py launches its two ranks itself, the ranks share
# the GPU, the full N > 1 line at 2^24 (tools/gpu.sh bench2)
export TAG=r06_e
OUT=gpurun_out
python bench.py --gpus 2 > $OUT/r06_e_gpus2_nccl_one_gpu.json 2> $OUT/r06_e_gpus2_nccl_one_gpu.err; echo "nccl on one GPU: rc=$?"; cat $OUT/r06_e_gpus2_nccl_one_gpu.json
WORLD_SIZE=1 RANK=0 python bench.py --gpus 2 > $OUT/r06_e_gpus2_world1.json 2>/dev/null; echo "WORLD_SIZE=1 --gpus 2: rc=$?"; cat $OUT/r06_e_gpus2_world1.json
tools/gpu.sh bench2
