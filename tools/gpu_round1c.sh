#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/bench_24.log 2>&1; echo "rc=$?" >> gpurun_out/bench_24.log
timeout 600 python bench.py --log-n 20 --no-cpu-baseline > gpurun_out/bench_20.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o b22 -- python bench.py --log-n 22 --steps 2 --warmup 1 --no-cpu-baseline --groth16-proofs 0 > gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o b22 -- python bench.py --log-n 22 --steps 2 --warmup 1 --no-cpu-baseline --groth16-proofs 0 > gpurun_out/prof_write.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
python - <<'PY'
import json
for f in ['gpurun_out/bench_24.log','gpurun_out/bench_20.log']:
    for line in open(f):
        if line.startswith('{'):
            d=json.loads(line); g=d.get('groth16',{})
            print(f, 'value', d['value'], 'ms/step', d['ms_per_step'], 'c', d['config']['window_bits'], 'roof', d['roofline']['frac'])
            print('  msm stages', {k:v['avg_ms'] for k,v in d['stages_ms'].items()})
            print('  groth16 ms', g.get('ms_per_proof'), 'computeH', g.get('computeH_ms'), {k:v['total_ms'] for k,v in g.get('stages_ms',{}).items()})
            print('  cpu', d.get('cpu_baseline'))
PY
