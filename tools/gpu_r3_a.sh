#!/bin/bash
# Round-3 batch A on one box: host facts, kernel A/B (NTT plans and builds, register accumulators, proof schedules), SQ counter passes
# of the new NTT pass and of the BLS12-381 bucket kernels, the bench line, the GPU suite.  Everything lands in gpurun_out/r3a/.
OUT=gpurun_out/r3a
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
{
  echo "nproc: $(nproc)"; echo "os.cpu_count / sched_getaffinity:"; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"
  echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
  lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)|MHz|NUMA"; free -g | head -2
  rocm-smi --showclocks 2>/dev/null | head -12
} > $OUT/host_info.txt 2>&1
run() { tag=$1; shift; timeout 300 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run base            $AB --parts ntt,msm --tag base
run plan888         GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag plan888
run plan1086        GA_NTT_PLAN=10,8,6 $AB --parts ntt --tag plan1086
run plan6666        GA_NTT_PLAN=6,6,6,6 $AB --parts ntt --tag plan6666
run gs              GA_LIB_PATH=$V/libgnark_amd_gs.so $AB --parts ntt --tag gs
run twu             GA_LIB_PATH=$V/libgnark_amd_twu.so $AB --parts ntt --tag twu
run twu888          GA_LIB_PATH=$V/libgnark_amd_twu.so GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag twu888
run accreg1         GA_LIB_PATH=$V/libgnark_amd_accreg1.so $AB --parts msm --tag accreg1
run g16             $AB --parts g16 --tag g16 --proofs 6
run bls_base        $AB --parts ntt,msm --curve bls12-381 --tag bls_base --reps 3
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3a/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    n = d.get("ntt", {}); m = d.get("msm", {}); g = d.get("g16", {})
    print(d["tag"], d["lib"], d["env"])
    if n: print("   ntt: ifft_dif %.3f fft_dit_coset %.3f ifft_dif_coset %.3f computeH %.3f  sha %s %s" % (n["ifft_dif_ms"], n["fft_dit_coset_ms"], n["ifft_dif_coset_ms"], n["compute_h_ms"], n["sha_ifft_dif"], n["sha_fft_dit_coset"]), n["passes"])
    for k, v in m.items(): print("   msm", k, v)
    if g: print("   g16", g)
P
cat $OUT/ab_summary.txt
# SQ counters: the NTT pass (new natural->bit-reversed form) and the BLS12-381 bucket kernels (missing in round 2)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_ntt$i -o sq -- python tools/ab_kernels.py --parts ntt --reps 1 > $OUT/sq_ntt$i.log 2>&1
  python tools/prof_summary.py --pmc $OUT/sq_ntt$i/sq_results.db 2>/dev/null | grep -E "ntt_pass|counter" | cut -c1-200 >> $OUT/sq_ntt_counters.txt
  rm -rf $OUT/sq_ntt$i
  if [ $i -le 2 ]; then
    timeout 400 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_bls$i -o sq -- python tools/ab_kernels.py --parts msm --curve bls12-381 --reps 1 > $OUT/sq_bls$i.log 2>&1
    python tools/prof_summary.py --pmc $OUT/sq_bls$i/sq_results.db 2>/dev/null | grep -E "accumulate29_kernel|reduce_groups29|counter" | cut -c1-200 >> $OUT/sq_bls_counters.txt
    rm -rf $OUT/sq_bls$i
  fi
done
python tools/sq_summary.py $OUT/sq_ntt_counters.txt > $OUT/sq_ntt_summary.txt 2>&1; cat $OUT/sq_ntt_summary.txt
python tools/sq_summary.py $OUT/sq_bls_counters.txt > $OUT/sq_bls_summary.txt 2>&1; cat $OUT/sq_bls_summary.txt
# the driver's bench line and the GPU suite on the same box
timeout 900 python bench.py > $OUT/bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 300 $OUT/bench_bn254.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r3a/bench_bn254_2p24.json").read().strip().splitlines()[-1])
    g = d["groth16"]
    print("bench:", d["value"], d["ms_per_step"], "g16", g["ms_per_proof"], g["schedule"]["split_proofs"], "profiled", g["ms_per_proof_profiled_single_lane"], "pipelined", g["pipelined"], "computeH", g["computeH_ms"], g.get("matches_dlog"), "plonk", d.get("plonk", {}).get("ms_per_proof_kernels"), d.get("plonk", {}).get("identity_ok"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench line unreadable:", e)
P
(time timeout 1200 python -m pytest tests -q -m gpu -x --durations=8) > $OUT/full_gpu_suite.log 2>&1; tail -15 $OUT/full_gpu_suite.log
