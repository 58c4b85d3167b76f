#!/bin/bash
# exercise the N>1 path of bench.py on a 1-GPU box: 2 ranks share device 0, gloo for the exchange
mkdir -p gpurun_out
export TMPDIR=/tmp
GA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --log-n ${LOGN:-20} > gpurun_out/bench_2rank_gloo.log 2>&1; echo "rc=$?" >> gpurun_out/bench_2rank_gloo.log
grep -E "^\{|rc=|Error|error" gpurun_out/bench_2rank_gloo.log | cut -c1-250
python - <<'PY'
import json
for line in open('gpurun_out/bench_2rank_gloo.log'):
    if line.startswith('{'):
        d=json.loads(line); print('value', d['value'], 'groth16', d.get('groth16'))
PY
timeout 600 python bench.py --log-n ${LOGN:-20} --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('single-rank groth16 sha', d['groth16']['proof_sha'], d['groth16']['ms_per_proof'])"
