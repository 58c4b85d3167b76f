#!/bin/bash
# first GPU visit: smoke, parity tests, microbench, small bench + rocprof
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "gfx|Marketing" > gpurun_out/rocminfo.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "
import gnark_amd, json
ctx = gnark_amd.Context(0)
print(json.dumps(ctx.info())); print(json.dumps(ctx.microbench()))
" > gpurun_out/microbench.log 2>&1
timeout 600 python bench.py --log-n 20 --steps 5 --warmup 1 > gpurun_out/bench_20.log 2>&1; echo "rc=$?" >> gpurun_out/bench_20.log
timeout 900 python bench.py --log-n 22 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_22.log 2>&1; echo "rc=$?" >> gpurun_out/bench_22.log
tail -3 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/microbench.log; tail -2 gpurun_out/bench_20.log | cut -c1-1500; tail -2 gpurun_out/bench_22.log | cut -c1-1500
