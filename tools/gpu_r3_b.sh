#!/bin/bash
# Round-3 batch B: dense twiddle tables (TS / TB) -- plans, LDS swizzle variant, proof schedules, SQ counters of the new pass
OUT=gpurun_out/r3b
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 300 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run base            $AB --parts ntt --tag base
run plan888         GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag plan888
run plan1086        GA_NTT_PLAN=10,8,6 $AB --parts ntt --tag plan1086
run plan8106        GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag plan888b
run swz             GA_LIB_PATH=$V/libgnark_amd_swz.so $AB --parts ntt --tag swz
run swz888          GA_LIB_PATH=$V/libgnark_amd_swz.so GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag swz888
run g16             $AB --parts g16 --tag g16 --proofs 6
run bls_base        $AB --parts ntt --curve bls12-381 --tag bls_base --reps 3
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b/ab_*.json")):
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
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_ntt$i -o sq -- python tools/ab_kernels.py --parts ntt --reps 1 > $OUT/sq_ntt$i.log 2>&1
  python tools/prof_summary.py --pmc $OUT/sq_ntt$i/sq_results.db 2>/dev/null | grep -E "ntt_pass|counter" | cut -c1-200 >> $OUT/sq_ntt_counters.txt
  rm -rf $OUT/sq_ntt$i
done
python tools/sq_summary.py $OUT/sq_ntt_counters.txt > $OUT/sq_ntt_summary.txt 2>&1; cat $OUT/sq_ntt_summary.txt
grep LDS $OUT/sq_ntt_counters.txt | cut -c1-120
(time timeout 900 python -m pytest tests -q -m gpu -x -k "fft or ntt or compute_h or plonk or groth16" ) > $OUT/gpu_subset.log 2>&1; tail -5 $OUT/gpu_subset.log
