#!/bin/bash
# round 2: the new ABI / multi-device paths on the 1-GPU box + the N>1 bench path with two ranks sharing the GPU (gloo)
mkdir -p gpurun_out
export TMPDIR=/tmp
(time python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staged or builder or prove_multi or window_ranges or cgo" --durations=8) > gpurun_out/r2_multi_tests.log 2>&1
tail -14 gpurun_out/r2_multi_tests.log
for part in range window; do
  GA_BENCH_BACKEND=gloo GA_BENCH_REPLICATE_H=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 2 --warmup 1 --log-n ${LOGN:-22} --groth16-proofs 2 --partition $part > gpurun_out/r2_bench_2rank_$part.log 2>&1
  echo "rc=$?"; grep -E "^\{" gpurun_out/r2_bench_2rank_$part.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$part', 'value', d['value'], 'groth16', d.get('groth16'))"
  grep -iE "error|Traceback" gpurun_out/r2_bench_2rank_$part.log | head -5
done
timeout 600 python bench.py --log-n ${LOGN:-22} --no-cpu-baseline --plonk-log-n 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('single-rank groth16', d['groth16']['proof_sha'], d['groth16']['ms_per_proof'], d['groth16'].get('matches_dlog'))"
