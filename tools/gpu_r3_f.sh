#!/bin/bash
# Round-3 batch F: twiddle prefetch across rounds (new build) vs the previous build (variants/libgnark_amd_nopf.so), plans 10,7,7 / 8,8,8
OUT=gpurun_out/r3f
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 400 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run pf              $AB --parts ntt --tag pf
run nopf            GA_LIB_PATH=$V/libgnark_amd_nopf.so $AB --parts ntt --tag nopf
run pf888           GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag pf888
run nopf888         GA_LIB_PATH=$V/libgnark_amd_nopf.so GA_NTT_PLAN=8,8,8 $AB --parts ntt --tag nopf888
run pf1086          GA_NTT_PLAN=10,8,6 $AB --parts ntt --tag pf1086
run pf_b            $AB --parts ntt,g16 --tag pf_b --proofs 6
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    n = d.get("ntt", {}); g = d.get("g16", {})
    print(d["tag"], d["lib"], d["env"])
    if n: print("   ntt: ifft_dif %.3f fft_dit_coset %.3f ifft_dif_coset %.3f computeH %.3f  sha %s %s" % (n["ifft_dif_ms"], n["fft_dit_coset_ms"], n["ifft_dif_coset_ms"], n["compute_h_ms"], n["sha_ifft_dif"], n["sha_fft_dit_coset"]), n["passes"])
    if g: print("   g16", g)
P
cat $OUT/ab_summary.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/sq_ntt$i -o sq -- python tools/ab_kernels.py --parts ntt --reps 1 > $OUT/sq_ntt$i.log 2>&1
  python tools/prof_summary.py --pmc $OUT/sq_ntt$i/sq_results.db 2>/dev/null | grep -E "ntt_pass|counter" | cut -c1-200 >> $OUT/sq_ntt_counters.txt
  rm -rf $OUT/sq_ntt$i
done
python tools/sq_summary.py $OUT/sq_ntt_counters.txt > $OUT/sq_ntt_summary.txt 2>&1; cat $OUT/sq_ntt_summary.txt
(time timeout 900 python -m pytest tests -q -m gpu -x -k "fft or ntt or compute_h or plonk" ) > $OUT/gpu_subset.log 2>&1; tail -3 $OUT/gpu_subset.log
