#!/bin/bash
# Round-3 batch X: radix-8 register blocks in the NTT pass (three stages per LDS round trip where possible: 10 = 3+3+2+2, 7 = 3+2+2;
# -DGA_NTT_RADIX8) against the shipped radix-4 rounds, same box; outputs hashed
OUT=gpurun_out/r3x
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 500 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 200 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py --parts ntt"
run base  $AB --tag base
run r8    GA_LIB_PATH=$V/libgnark_amd_r8.so $AB --tag r8
run base2 $AB --tag base2
run r8b   GA_LIB_PATH=$V/libgnark_amd_r8.so $AB --tag r8b
run r8_888   GA_LIB_PATH=$V/libgnark_amd_r8.so GA_NTT_PLAN=9,9,6 $AB --tag r8_996
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3x/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    n = d.get("ntt", {})
    print(d["tag"], d["lib"], d["env"])
    if n: print("   ntt: ifft_dif %.3f fft_dit_coset %.3f ifft_dif_coset %.3f computeH %.3f  sha %s %s" % (n["ifft_dif_ms"], n["fft_dit_coset_ms"], n["ifft_dif_coset_ms"], n["compute_h_ms"], n["sha_ifft_dif"], n["sha_fft_dit_coset"]), n["passes"])
P
cat $OUT/ab_summary.txt; cat $OUT/failures.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/st -o k -- env GA_LIB_PATH=$V/libgnark_amd_r8.so python tools/ab_kernels.py --parts ntt --reps 2 > /dev/null 2>&1
python tools/prof_summary.py $OUT/st/k_results.db 2>/dev/null | grep -E "calls|ntt_pass" | cut -c1-150 > $OUT/r8_kernel_stats.txt; rm -rf $OUT/st; cat $OUT/r8_kernel_stats.txt
