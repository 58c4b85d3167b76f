#!/bin/bash
# Round-2 closing run on one box: full GPU suite, smoke(), the plain bench lines (both curves) and the kernel-stats pass.
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
(time python -m pytest tests -q -m gpu --durations=10) > $OUT/r02_e_full_gpu_suite.log 2>&1; tail -5 $OUT/r02_e_full_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py > $OUT/r02_e_bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 600 $OUT/bench_bn254.err
python bench.py --curve bls12-381 --plonk-log-n 0 > $OUT/r02_e_bench_bls12381_2p24.json 2> $OUT/bench_bls.err
python tools/bench_plonk_kernels.py > $OUT/r02_e_bench_plonk_2p22.json 2>/dev/null
for c in bn254 bls12-381; do
  cc=${c//-/}
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${cc}_stats -o k -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0 --curve $c > $OUT/${cc}_stats.log 2>&1
  python tools/prof_summary.py $OUT/${cc}_stats/k_results.db > $OUT/r02_e_bench24_${cc}_kernel_stats.txt 2>/dev/null
  rm -rf $OUT/${cc}_stats
done
python - <<'P'
import json
for c in ("bn254", "bls12381"):
    d = json.load(open("gpurun_out/final/r02_e_bench_%s_2p24.json" % c))
    g = d["groth16"]
    print(c, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], g["ms_per_proof"], g["pipelined"]["ms_per_proof"], g.get("matches_dlog"), d.get("plonk", {}).get("identity_ok"), d.get("cpu_baseline", {}).get("value"))
P
head -14 $OUT/r02_e_bench24_bn254_kernel_stats.txt | cut -c1-170
