#!/bin/bash
# Round-3 batch Y: the NTT pass before / after its round loop was restructured (default schedule), same box
OUT=gpurun_out/r3y
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/gnark_amd/variants
run() { tag=$1; shift; timeout 500 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; }
AB="python tools/ab_kernels.py --parts ntt"
run new1  $AB --tag new1
run old1  GA_LIB_PATH=$V/libgnark_amd_oldntt.so $AB --tag old1
run new2  $AB --tag new2
run old2  GA_LIB_PATH=$V/libgnark_amd_oldntt.so $AB --tag old2
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3y/ab_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); n = d["ntt"]
    print(d["tag"], d["lib"], "ifft_dif %.3f fft_dit_coset %.3f ifft_dif_coset %.3f computeH %.3f" % (n["ifft_dif_ms"], n["fft_dit_coset_ms"], n["ifft_dif_coset_ms"], n["compute_h_ms"]), n["sha_fft_dit_coset"])
P
