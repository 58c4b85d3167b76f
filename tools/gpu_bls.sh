#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 1500 python bench.py --curve bls12-381 --no-cpu-baseline > gpurun_out/bench_bls_24.json 2> gpurun_out/bench_bls_24.err
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/bench_bn254_24.json 2> gpurun_out/bench_bn254_24.err
timeout 900 python tools/bench_plonk_kernels.py > gpurun_out/bench_plonk_22.json 2> gpurun_out/bench_plonk_22.err
tail -2 gpurun_out/pytest_gpu.log
for f in bench_bn254_24 bench_bls_24 bench_plonk_22; do echo "== $f"; python - <<PY
import json
for line in open('gpurun_out/$f.json'):
    if line.startswith('{'):
        d=json.loads(line); g=d.get('groth16',{})
        print({k:d[k] for k in ('value','ms_per_step','ms_per_proof_kernels','msm_ms','ntt_ms','hbm_frac','ntt_hbm_frac') if k in d}, 'groth16', g.get('ms_per_proof'), g.get('proofs_per_s'), 'setup', g.get('key_setup_s'))
        print('   ', {k:(v['total_ms'] if isinstance(v,dict) else v) for k,v in (g.get('stages_ms') or d.get('stages_ms')).items()})
PY
done
