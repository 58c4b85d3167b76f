#!/bin/bash
# Round-3 batch E: same-box A/B of the coset-folded twiddle tables (GA_NTT_COSET_FOLD=0/1), two-caller variance, the bench line
OUT=gpurun_out/r3e
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 400 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run fold1           $AB --parts ntt,g16 --tag fold1 --proofs 8
run fold0           GA_NTT_COSET_FOLD=0 $AB --parts ntt,g16 --tag fold0 --proofs 8
run fold1b          $AB --parts ntt,g16 --tag fold1b --proofs 8
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3e/ab_*.json")):
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
for v in 1 0; do
GA_NTT_COSET_FOLD=$v timeout 600 python tools/bench_plonk_kernels.py 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('plonk fold=$v', d['ms_per_proof_kernels'], d['ntt_ms'], d['msm_ms'])"
done
timeout 900 python bench.py > $OUT/bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 300 $OUT/bench_bn254.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r3e/bench_bn254_2p24.json").read().strip().splitlines()[-1])
    g = d["groth16"]
    print("bench:", d["value"], d["value_checked"], d["ms_per_step"], "g16", g["ms_per_proof"], "profiled", g["ms_per_proof_profiled_single_lane"], "pipelined", g["pipelined"]["ms_per_proof"], g["pipelined"]["vs_single_caller"], g["pipelined"]["lanes"], "computeH", g["computeH_ms"], g.get("matches_dlog"), "plonk", d.get("plonk", {}).get("ms_per_proof_kernels"), d.get("plonk", {}).get("identity_ok"))
    print({k: v["total_ms"] for k, v in g["stages_ms"].items()})
except Exception as e:
    print("bench line unreadable:", e)
P
