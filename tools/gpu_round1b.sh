#!/bin/bash
# full-size bench + rocprofv3 kernel stats + PMC HBM traffic
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
timeout 1500 python bench.py > gpurun_out/bench_24.log 2>&1; echo "rc=$?" >> gpurun_out/bench_24.log
timeout 1500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o bench24 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_stats.log 2>&1; echo "rc=$?" >> gpurun_out/prof_stats.log
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o b22 -- python bench.py --log-n 22 --steps 2 --warmup 1 --no-cpu-baseline --groth16-proofs 0 > gpurun_out/prof_fetch.log 2>&1; echo "rc=$?" >> gpurun_out/prof_fetch.log
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o b22 -- python bench.py --log-n 22 --steps 2 --warmup 1 --no-cpu-baseline --groth16-proofs 0 > gpurun_out/prof_write.log 2>&1; echo "rc=$?" >> gpurun_out/prof_write.log
find gpurun_out -name "*.csv" | head -20
tail -2 gpurun_out/bench_24.log | cut -c1-3000
