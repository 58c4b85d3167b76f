#!/bin/bash
# Round-3 batch V: workgroup size of the bucket kernels of the 14-limb fields (BLS12-381): 256 / 128 lanes (shipped) vs 128 vs 64 --
# more waves per CU under the 160 KB of LDS (G1 8 -> 10 -> 11, G2 4 -> 5)
OUT=gpurun_out/r3v
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 600 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 200 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py --curve bls12-381 --parts msm,g16 --proofs 4"
run base   $AB --tag base
run wg64   GA_LIB_PATH=$V/libgnark_amd_wg64.so $AB --tag wg64
run wg128  GA_LIB_PATH=$V/libgnark_amd_wg128.so $AB --tag wg128
run base2  $AB --tag base2
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3v/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    m = d.get("msm", {}); g = d.get("g16", {})
    print(d["tag"], d["lib"])
    for k in ("g1", "g2"):
        if k in m: print("   ", k, m[k])
    print("    g16", {k: g.get(k) for k in ("split_ms", "split_two_callers_ms", "single_lane_ms", "split_sha")})
P
cat $OUT/ab_summary.txt; cat $OUT/failures.txt 2>/dev/null
