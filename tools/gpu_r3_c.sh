#!/bin/bash
# Round-3 batch C: the G1 bucket kernel capped at 3 waves per SIMD (room for co-resident kernels of the partner lane) vs 4;
# the reworked bench line; the full GPU suite
OUT=gpurun_out/r3c
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 400 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run base            $AB --parts msm,g16 --tag base --proofs 8
run w3              GA_LIB_PATH=$V/libgnark_amd_w3.so $AB --parts msm,g16 --tag w3 --proofs 8
run base2           $AB --parts g16 --tag base2 --proofs 8
run w3b             GA_LIB_PATH=$V/libgnark_amd_w3.so $AB --parts g16 --tag w3b --proofs 8
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    m = d.get("msm", {}); g = d.get("g16", {})
    print(d["tag"], d["lib"], d["env"])
    for k, v in m.items(): print("   msm", k, v)
    if g: print("   g16", g)
P
cat $OUT/ab_summary.txt
timeout 900 python bench.py > $OUT/bench_bn254_2p24.json 2> $OUT/bench_bn254.err; tail -c 300 $OUT/bench_bn254.err
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r3c/bench_bn254_2p24.json").read().strip().splitlines()[-1])
    g = d["groth16"]
    print("bench:", d["value"], d["value_checked"], d["ms_per_step"], "g16", g["ms_per_proof"], g["schedule"]["split_proofs"], "profiled", g["ms_per_proof_profiled_single_lane"], "pipelined", g["pipelined"]["ms_per_proof"], g["pipelined"]["vs_single_caller"], g["pipelined"]["lanes"], "computeH", g["computeH_ms"], g.get("matches_dlog"), "plonk", d.get("plonk", {}).get("ms_per_proof_kernels"), d.get("plonk", {}).get("identity_ok"))
    print("cpu", d.get("cpu_baseline"))
    print("config2", d.get("config2_msm_2p20_unpinned"))
except Exception as e:
    print("bench line unreadable:", e)
P
(time timeout 1200 python -m pytest tests -q -m gpu -x --durations=5) > $OUT/full_gpu_suite.log 2>&1; tail -12 $OUT/full_gpu_suite.log
