#!/bin/bash
# round 5, batch W: what the round changed, box spread taken out -- `python bench.py` (counter passes and RCCL self-test off) on ONE box,
# first with round 4's library (commit ed12e17's gnark_amd/csrc built as gnark_amd/variants/libgnark_amd_r04.so; same 84 exported
# symbols, GA_LIB_PATH), then with the final one, then round 4's again
export TMPDIR=/tmp
OUT=gpurun_out
V=/root/repo/gnark_amd/variants/libgnark_amd_r04.so
for tag in r04 r05 r04b; do
  lib=""; [ "$tag" != "r05" ] && lib="GA_LIB_PATH=$V"
  echo "=== $tag"
  env $lib timeout 600 python bench.py --no-pmc --no-selftest --detail-file $OUT/r05_w_bench_${tag}_detail.json > $OUT/r05_w_bench_$tag.json 2> $OUT/r05_w_bench_$tag.err
  tail -2 $OUT/r05_w_bench_$tag.err
  python tools/bench_digest.py $OUT/r05_w_bench_$tag.json | grep -E "headline|stages|msm_ms|groth16_bn254_ms|two_callers_ms|computeH|bls12_381_ms|plonk_bn254|plain_msm|h2d|config2|g1_ms|g2_ms"
done
