#!/bin/bash
# bench at 2^24 (+ Groth16 leg) and 2^20, no tests
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python bench.py --no-cpu-baseline $BENCH_ARGS > gpurun_out/bench_24.log 2>&1; echo "rc=$?" >> gpurun_out/bench_24.log
timeout 600 python bench.py --log-n 20 --no-cpu-baseline > gpurun_out/bench_20.log 2>&1
python - <<'PY'
import json
for f in ['gpurun_out/bench_24.log','gpurun_out/bench_20.log']:
    for line in open(f):
        if line.startswith('{'):
            d=json.loads(line); g=d.get('groth16',{})
            print(f, 'value', d['value'], 'ms/step', d['ms_per_step'], 'c', d['config']['window_bits'])
            print('  msm stages', {k:v['avg_ms'] for k,v in d['stages_ms'].items()})
            print('  groth16 ms', g.get('ms_per_proof'), 'computeH', g.get('computeH_ms'), {k:v['total_ms'] for k,v in g.get('stages_ms',{}).items()})
        elif 'rror' in line: print(line[:300])
PY
