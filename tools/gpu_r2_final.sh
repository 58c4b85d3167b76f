#!/bin/bash
# Round-2 closing run on one box: full GPU suite, smoke(), the plain bench lines (both curves), the PLONK leg and the kernel-stats
# passes (two-caller leg excluded from profiler runs).  TAG names the output files.
TAG=${TAG:-r02_f}
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
(time python -m pytest tests -q -m gpu --durations=10) > $OUT/${TAG}_full_gpu_suite.log 2>&1; tail -5 $OUT/${TAG}_full_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py > $OUT/${TAG}_bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 300 $OUT/bench_bn254.err
python bench.py --curve bls12-381 --plonk-log-n 0 > $OUT/${TAG}_bench_bls12381_2p24.json 2> $OUT/bench_bls.err
python tools/bench_plonk_kernels.py > $OUT/${TAG}_bench_plonk_2p22.json 2>/dev/null
for c in bn254 bls12-381; do
  cc=${c//-/}
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${cc}_stats -o k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0 --curve $c > $OUT/${cc}_stats.log 2>&1
  python tools/prof_summary.py $OUT/${cc}_stats/k_results.db > $OUT/${TAG}_bench24_${cc}_kernel_stats.txt 2>/dev/null
  rm -rf $OUT/${cc}_stats
done
python - <<P
import json
for c in ("bn254", "bls12381"):
    d = json.load(open("gpurun_out/final/${TAG}_bench_%s_2p24.json" % c))
    g = d["groth16"]
    print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_ms"], g["ms_per_proof"], g["pipelined"]["ms_per_proof"], g.get("matches_dlog"), d.get("plonk", {}).get("identity_ok"), d.get("plonk", {}).get("ms_per_proof_kernels"), d.get("cpu_baseline", {}).get("value"))
P
grep -E "accumulate29_kernel|radix_sort|table29" $OUT/${TAG}_bench24_bn254_kernel_stats.txt | head -8 | cut -c1-170
