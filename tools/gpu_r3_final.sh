#!/bin/bash
# Round-3 closing run on one box: full GPU suite, smoke(), the bench lines (both curves), the PLONK leg, kernel stats, FETCH/WRITE and
# SQ passes of the final code.  TAG names the output files (copied into profiles/ afterwards).
TAG=${TAG:-r03_final}
OUT=gpurun_out/final3
mkdir -p $OUT
export TMPDIR=/tmp
(time python -m pytest tests -q -m gpu --durations=8) > $OUT/${TAG}_full_gpu_suite.log 2>&1; tail -4 $OUT/${TAG}_full_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
python bench.py > $OUT/${TAG}_bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 300 $OUT/bench_bn254.err
python bench.py --curve bls12-381 --plonk-log-n 0 > $OUT/${TAG}_bench_bls12381_2p24.json 2> $OUT/bench_bls.err
python tools/bench_plonk_kernels.py > $OUT/${TAG}_bench_plonk_2p22.json 2>/dev/null
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0"
for c in bn254 bls12-381; do
  cc=${c//-/}
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${cc}_stats -o k -- $BENCH --curve $c > $OUT/${cc}_stats.log 2>&1
  python tools/prof_summary.py $OUT/${cc}_stats/k_results.db > $OUT/${TAG}_bench24_${cc}_kernel_stats.txt 2>/dev/null
  rm -rf $OUT/${cc}_stats
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/bn254_$ctr -o p -- $BENCH > $OUT/bn254_$ctr.log 2>&1 || echo "rocprofv3 $ctr failed"
  python tools/prof_summary.py --pmc $OUT/bn254_$ctr/p_results.db > $OUT/${TAG}_bench24_bn254_pmc_${ctr}.txt
  rm -rf $OUT/bn254_$ctr
done
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq$i -o sq -- $BENCH > $OUT/sq$i.log 2>&1
  python tools/prof_summary.py --pmc $OUT/sq$i/sq_results.db 2>/dev/null | grep -E "counter|accumulate29_kernel|reduce_groups29|ntt_pass29r4|radix_sort_onesweep" | cut -c1-220 >> $OUT/${TAG}_sq_counters.txt
  rm -rf $OUT/sq$i
done
python tools/sq_summary.py $OUT/${TAG}_sq_counters.txt > $OUT/${TAG}_sq_summary.txt 2>&1; cat $OUT/${TAG}_sq_summary.txt
python - <<P
import json
for c in ("bn254", "bls12381"):
    d = json.loads(open("gpurun_out/final3/${TAG}_bench_%s_2p24.json" % c).read().strip().splitlines()[-1])
    g = d["groth16"]
    print(c, d["value"], d.get("value_checked"), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "g16", g["ms_per_proof"], "pipe", g["pipelined"]["ms_per_proof"], g["pipelined"]["vs_single_caller"], g.get("matches_dlog"), "computeH", g["computeH_ms"], "plonk", d.get("plonk", {}).get("ms_per_proof_kernels"), d.get("plonk", {}).get("identity_ok"), "cpu", d.get("cpu_baseline", {}).get("value"))
P
grep -E "accumulate29_kernel|radix_sort|ntt_pass" $OUT/${TAG}_bench24_bn254_kernel_stats.txt | head -8 | cut -c1-170
grep -E "ntt_pass|accumulate29" $OUT/${TAG}_bench24_bn254_pmc_FETCH_SIZE.txt | head -4 | cut -c1-170
# the --gpus N code path on this 1-GPU box: two ranks share the GPU, collectives over gloo (control flow + self-checks; timings mean nothing)
GA_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 > $OUT/${TAG}_bench_2ranks_one_gpu_gloo.json 2> $OUT/bench_2ranks.err
python - <<P
import json
d = json.loads(open("gpurun_out/final3/${TAG}_bench_2ranks_one_gpu_gloo.json").read().strip().splitlines()[-1])
print("2 ranks: value_checked", d.get("value_checked"), "groth16", {k: d["groth16"].get(k) for k in ("ms_per_proof", "matches_dlog", "proof_sha", "error")})
P
