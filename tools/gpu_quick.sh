#!/bin/bash
# quick GPU check: parity tests + microbench + bench at 2^24 and 2^20 (no rocprof)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "
import gnark_amd, json
ctx = gnark_amd.Context(0)
print(json.dumps(ctx.microbench()))
" > gpurun_out/microbench.log 2>&1
timeout 1500 python bench.py $BENCH_ARGS > gpurun_out/bench_24.log 2>&1; echo "rc=$?" >> gpurun_out/bench_24.log
timeout 600 python bench.py --log-n 20 --no-cpu-baseline > gpurun_out/bench_20.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/microbench.log
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
        elif 'rror' in line: print(line[:300])
PY
