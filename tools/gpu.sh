#!/bin/bash
# The one parameterised driver for everything that runs on the GPU box (the per-batch gpu_r2_* / gpu_r3_* / gpu_<topic> scripts of the
# earlier rounds are gone; their outputs under profiles/ name the steps below).
#   TAG=r04_a tools/gpu.sh <step> [<step> ...]        steps run in order; outputs go to gpurun_out/${TAG}_*
# steps:
#   tests[:<pytest -k expression>]   the -m gpu suite (or a selection)
#   smoke                            __graft_entry__.smoke()
#   micro                            ga_microbench: instruction issue rates (burst and sustained), dependency distance x occupancy
#   bench[:<extra bench.py args>]    python bench.py -> ${TAG}_bench.json  (":--curve bls12-381" etc.; spaces as '+')
#   bench2                           `python bench.py --gpus 2` as the driver would type it (bench.py launches its ranks): two ranks share this box's GPU, collectives over gloo
#   stats[:<bench args>]             rocprofv3 --kernel-trace --stats of the headline leg (bench.py --only-headline) -> ${TAG}_kernel_stats.txt
#   stats_proof                      the same over a short bench WITH a Groth16 proof (multi-table launches included) -> ${TAG}_kernel_stats_proof.txt
#   hbm[:<bench args>]               FETCH_SIZE / WRITE_SIZE passes of the headline leg (separate, kernel-trace only) -> ${TAG}_pmc_{FETCH,WRITE}_SIZE.txt
#   sq:<name>:<driver>               the SQ issue accounting + the wait split (LDS / VMEM / instruction cache) of <driver>:
#                                    bench | bls (BLS12-381 G1+G2 table MSMs) | bn (BN254 ones) | ntt  -> ${TAG}_sq_<name>_{counters,summary}.txt
#   ab:<name>:<parts>[:<curve>[:<variant>]]   tools/ab_kernels.py --parts <parts>, on gnark_amd/variants/libgnark_amd_<variant>.so when given
#                                    (run-time knobs from the environment) -> ${TAG}_ab_<name>.json
#   abenv:<name>:<parts>:<curve>:<K=V,K=V,...>   the same with run-time knobs set for that one run ("-" = none): same-box A/B of a knob
#                                    (appends to ${TAG}_abenv.txt, one JSON line per run)
#   g16ab:<name>:<KNOB=v1,v2>[:<curve>[:<log-n>]]   same-box, same-process A/B of a run-time knob on one pinned Groth16 key (settings
#                                    interleaved; one- and two-caller proof times per round + the profiled stage table) -> ${TAG}_g16ab_<name>.json
#   plonk                            tools/bench_plonk_kernels.py (config 5 kernel work) -> ${TAG}_bench_plonk.json
#   clock                            tools/clock_probe.py: shader clock / power held under each kernel family -> ${TAG}_clock_probe.json
#   small:<log-n>                    tools/msm_small_trace.py: stage profile of a raw and a table MSM of 2^<log-n> points
# Counter passes never combine --pmc with anything but --kernel-trace.
TAG=${TAG:-r04}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
SHORT_BENCH="--steps 3 --warmup 1 --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0 --no-bls --no-selftest --no-pmc"

driver_cmd() {
  case "$1" in
    bench) echo "python bench.py $SHORT_BENCH" ;;
    bls) echo "python tools/ab_kernels.py --parts msm --curve bls12-381 --reps 2" ;;
    bn) echo "python tools/ab_kernels.py --parts msm --curve bn254 --reps 2" ;;
    ntt) echo "python tools/ab_kernels.py --parts ntt --reps 2" ;;
    *) echo "$1" ;;
  esac
}

pmc_pass() {   # pmc_pass <outfile> <counters...> -- <cmd>
  local outfile=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  local d=$OUT/pmc_tmp_$$
  rm -rf $d
  timeout 900 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d $d -o p -- "$@" > $OUT/pmc_last.log 2>&1 || echo "rocprofv3 ${ctrs[*]} failed (see $OUT/pmc_last.log)"
  python tools/prof_summary.py --pmc $d/p_results.db 2>/dev/null | grep -E "counter|accumulate29_kernel|reduce_groups29|ntt_pass29r4|msm_p2_|msm_digits_pass1|plonk_" | cut -c1-230 >> $outfile
  rm -rf $d
}

for step in "$@"; do
  name=${step%%:*}
  arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  arg=${arg//+/ }
  echo "=== $step"
  case "$name" in
    tests)
      if [ -n "$arg" ]; then
        (time timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -k "$arg" --durations=5) > $OUT/${TAG}_gpu_tests_selected.log 2>&1
        tail -5 $OUT/${TAG}_gpu_tests_selected.log
      else
        (time timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8) > $OUT/${TAG}_full_gpu_suite.log 2>&1
        tail -5 $OUT/${TAG}_full_gpu_suite.log
      fi ;;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2 ;;
    micro)
      python -c "import gnark_amd, json; ctx = gnark_amd.Context(0); print(json.dumps(ctx.microbench()))" > $OUT/${TAG}_microbench.json 2> $OUT/${TAG}_microbench.err
      tail -2 $OUT/${TAG}_microbench.err; cat $OUT/${TAG}_microbench.json ;;
    bench)
      sfx=$(echo "$arg" | tr -cd 'a-z0-9' | cut -c1-24)
      (time timeout 2400 python bench.py --detail-file $OUT/${TAG}_bench${sfx:+_$sfx}_detail.json $arg) > $OUT/${TAG}_bench${sfx:+_$sfx}.json 2> $OUT/${TAG}_bench${sfx:+_$sfx}.err
      tail -4 $OUT/${TAG}_bench${sfx:+_$sfx}.err
      python tools/bench_digest.py $OUT/${TAG}_bench${sfx:+_$sfx}.json ;;
    bench2)
      # the PLAIN form -- the shape of the driver's BENCH command: no launcher, bench.py starts its two ranks itself (round 6)
      GA_BENCH_BACKEND=gloo timeout 1500 python bench.py --gpus 2 --steps 3 --warmup 1 --detail-file $OUT/${TAG}_bench_2ranks_one_gpu_gloo_detail.json $arg > $OUT/${TAG}_bench_2ranks_one_gpu_gloo.json 2> $OUT/${TAG}_bench_2ranks.err
      tail -3 $OUT/${TAG}_bench_2ranks.err
      python tools/bench_digest.py $OUT/${TAG}_bench_2ranks_one_gpu_gloo.json ;;
    stats)
      d=$OUT/stats_tmp_$$
      # the HEADLINE leg alone (20 MSM steps): since round 6 a proof launches the bucket kernel once for THREE tables (A, B1, K), so a
      # profile of the whole short bench would average single- and triple-table launches under one kernel name
      timeout 900 rocprofv3 --kernel-trace --stats -d $d -o k -- python bench.py --only-headline --steps 20 --warmup 2 ${arg:-} > $OUT/${TAG}_stats.log 2>&1
      python tools/prof_summary.py $d/k_results.db > $OUT/${TAG}_kernel_stats.txt 2>/dev/null
      rm -rf $d
      head -12 $OUT/${TAG}_kernel_stats.txt | cut -c1-200 ;;
    stats_proof)
      d=$OUT/stats_tmp_$$
      timeout 900 rocprofv3 --kernel-trace --stats -d $d -o k -- python bench.py $SHORT_BENCH $arg > $OUT/${TAG}_stats_proof.log 2>&1
      python tools/prof_summary.py $d/k_results.db > $OUT/${TAG}_kernel_stats_proof.txt 2>/dev/null
      rm -rf $d
      head -10 $OUT/${TAG}_kernel_stats_proof.txt | cut -c1-200 ;;
    hbm)
      for ctr in FETCH_SIZE WRITE_SIZE; do
        rm -f $OUT/${TAG}_pmc_${ctr}.txt
        pmc_pass $OUT/${TAG}_pmc_${ctr}.txt $ctr -- python bench.py --only-headline --steps 10 --warmup 2 ${arg:-}
        head -5 $OUT/${TAG}_pmc_${ctr}.txt | cut -c1-170
      done ;;
    sq)
      nm=${arg%%:*}; drv=${arg#*:}
      cmd=$(driver_cmd "$drv")
      f=$OUT/${TAG}_sq_${nm}_counters.txt
      rm -f $f
      for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
                 "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
                 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
        pmc_pass $f $set -- $cmd
      done
      python tools/sq_summary.py $f > $OUT/${TAG}_sq_${nm}_summary.txt 2>&1
      cat $OUT/${TAG}_sq_${nm}_summary.txt ;;
    ab)
      IFS=: read -r nm parts curve variant <<< "$arg"
      libenv=""; [ -n "$variant" ] && libenv="GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_${variant}.so"
      env $libenv timeout 1200 python tools/ab_kernels.py --parts $parts --curve ${curve:-bn254} --tag $nm > $OUT/${TAG}_ab_${nm}.json 2> $OUT/${TAG}_ab_${nm}.err
      tail -2 $OUT/${TAG}_ab_${nm}.err; cut -c1-1500 $OUT/${TAG}_ab_${nm}.json ;;
    abenv)
      IFS=: read -r nm parts curve kv <<< "$arg"
      envs=""; [ -n "$kv" ] && [ "$kv" != "-" ] && envs=${kv//,/ }
      env $envs timeout 1200 python tools/ab_kernels.py --parts $parts --curve ${curve:-bn254} --tag "$nm" >> $OUT/${TAG}_abenv.txt 2>> $OUT/${TAG}_abenv.err
      tail -1 $OUT/${TAG}_abenv.txt | cut -c1-900 ;;
    g16ab)   # g16ab:<name>:<KNOB=v1,v2>[:<curve>[:<log-n>]]  one key, one process: the knob's settings interleaved round by round
      IFS=: read -r nm knob curve logn <<< "$arg"
      timeout 1500 python tools/ab_kernels.py --parts g16ab --knob "$knob" --curve ${curve:-bn254} --log-n ${logn:-24} --tag "$nm" > $OUT/${TAG}_g16ab_${nm}.json 2> $OUT/${TAG}_g16ab_${nm}.err
      tail -2 $OUT/${TAG}_g16ab_${nm}.err; cut -c1-2500 $OUT/${TAG}_g16ab_${nm}.json ;;
    plonk)
      timeout 900 python tools/bench_plonk_kernels.py > $OUT/${TAG}_bench_plonk.json 2> $OUT/${TAG}_bench_plonk.err
      cut -c1-1200 $OUT/${TAG}_bench_plonk.json ;;
    clock)
      timeout 600 python tools/clock_probe.py --seconds 2.5 > $OUT/${TAG}_clock_probe.json 2> $OUT/${TAG}_clock_probe.err
      cut -c1-2500 $OUT/${TAG}_clock_probe.json ;;
    small)
      timeout 600 python tools/msm_small_trace.py --log-n ${arg:-20} --reps 20 > $OUT/${TAG}_msm_2p${arg:-20}_trace.json 2> $OUT/${TAG}_msm_small.err
      cat $OUT/${TAG}_msm_2p${arg:-20}_trace.json ;;
    *) echo "unknown step $step" ;;
  esac
done
