#!/bin/bash
# Round-2 rocprofv3 evidence, one gpurun call: kernel stats, FETCH/WRITE passes (separate), SQ counter passes, for BN254 and
# BLS12-381 at 2^24 and the PLONK leg.  Summaries are written as text under gpurun_out/prof_r2/ (copy into profiles/).
#   TAG=r02_d PASSES="stats pmc sq" tools/gpu_r2_profiles.sh
TAG=${TAG:-r02}
PASSES=${PASSES:-"stats pmc sq"}
OUT=gpurun_out/prof_r2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined"
run() {  # name, rocprof args..., -- cmd
  name=$1; shift
  timeout 900 rocprofv3 "$@" > $OUT/$name.log 2>&1 || echo "rocprofv3 $name failed (rc=$?)"
}
for curve in bn254 bls12-381; do
  c=${curve//-/}
  plonk="--plonk-log-n 0"; [ $curve = bn254 ] && plonk=""
  if [[ $PASSES == *stats* ]]; then
    run ${c}_stats --kernel-trace --stats -d $OUT/${c}_stats -o k -- $BENCH --curve $curve $plonk
    python tools/prof_summary.py $OUT/${c}_stats/k_results.db > $OUT/${TAG}_bench24_${c}_kernel_stats.txt 2>/dev/null
    head -16 $OUT/${TAG}_bench24_${c}_kernel_stats.txt | cut -c1-200
  fi
  if [[ $PASSES == *pmc* ]]; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
      run ${c}_$ctr --pmc $ctr --kernel-trace -d $OUT/${c}_$ctr -o p -- $BENCH --curve $curve $plonk
      python tools/prof_summary.py --pmc $OUT/${c}_$ctr/p_results.db > $OUT/${TAG}_bench24_${c}_pmc_${ctr}.txt 2>/dev/null
      head -8 $OUT/${TAG}_bench24_${c}_pmc_${ctr}.txt | cut -c1-200
    done
  fi
done
if [[ $PASSES == *sq* ]]; then
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
    i=$((i+1))
    run sq$i --pmc $set --kernel-trace -d $OUT/sq$i -o sq -- $BENCH --plonk-log-n 0
    python tools/prof_summary.py --pmc $OUT/sq$i/sq_results.db 2>/dev/null | grep -E "counter|accumulate29_kernel|reduce_groups29|ntt_pass29r4|radix_sort_onesweep" | cut -c1-220 >> $OUT/${TAG}_sq_counters.txt
  done
  cat $OUT/${TAG}_sq_counters.txt
fi
rm -rf $OUT/*_stats $OUT/*_FETCH_SIZE $OUT/*_WRITE_SIZE $OUT/sq1 $OUT/sq2 $OUT/sq3   # keep the text summaries only (64 MiB merge limit)
# plain bench lines for both curves (no profiler attached), for profiles/
if [[ $PASSES == *bench* ]]; then
  python bench.py > $OUT/${TAG}_bench_bn254_2p24.json 2> $OUT/bench_bn254.err
  python bench.py --curve bls12-381 --plonk-log-n 0 > $OUT/${TAG}_bench_bls12381_2p24.json 2> $OUT/bench_bls.err
  python - <<PY
import json
for c in ("bn254", "bls12381"):
    d = json.load(open("$OUT/${TAG}_bench_%s_2p24.json" % c))
    g = d["groth16"]
    print(c, "msm", d["ms_per_step"], d["value"], "acc", d["stages_ms"]["msm_accumulate"]["avg_ms"], "| g16", g["ms_per_proof"], g["proofs_per_s"], g.get("matches_dlog"), "pipelined", g["pipelined"]["ms_per_proof"], g["pipelined"]["proofs_per_s"], "| cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("groth16"))
PY
fi
