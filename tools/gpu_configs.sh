#!/bin/bash
# BASELINE configs 3 (BN254), 4 (BLS12-381, single-GPU leg) and 5 (PLONK kernels) + rocprofv3 kernel stats and PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 1500 python bench.py > gpurun_out/bench_bn254_24.json 2> gpurun_out/bench_bn254_24.err
timeout 1500 python bench.py --curve bls12-381 > gpurun_out/bench_bls_24.json 2> gpurun_out/bench_bls_24.err
timeout 900 python tools/bench_plonk_kernels.py > gpurun_out/bench_plonk_22.json 2> gpurun_out/bench_plonk_22.err
timeout 300 python -c "
import gnark_amd, json
ctx = gnark_amd.Context(0)
print(json.dumps(ctx.microbench()))
" > gpurun_out/microbench.json 2>&1
timeout 1500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bench24 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --plonk-log-n 0 > gpurun_out/prof_stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o b24 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --plonk-log-n 0 > gpurun_out/prof_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o b24 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --plonk-log-n 0 > gpurun_out/prof_write.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
for f in bench_bn254_24 bench_bls_24 bench_plonk_22; do echo "== $f"; tail -c 400 gpurun_out/$f.err; python - <<PY
import json
for line in open('gpurun_out/$f.json'):
    if line.startswith('{'):
        d=json.loads(line); g=d.get('groth16',{})
        print({k:d[k] for k in ('value','ms_per_step','ms_per_proof_kernels','msm_ms','ntt_ms','hbm_frac','ntt_hbm_frac') if k in d}, 'groth16', g.get('ms_per_proof'), g.get('proofs_per_s'))
        print('   ', {k:(v['total_ms'] if isinstance(v,dict) else v) for k,v in (g.get('stages_ms') or d.get('stages_ms')).items()})
        print('   cpu', d.get('cpu_baseline'))
PY
done
