#!/bin/bash
# Round-3 batch ZZ2: digits fused with the first radix-sort pass (msm.hip.h 1b) -- parity tests, then A/B against the plain
# digits + two-pass sort sequence (GA_MSM_FUSE_MIN above every size) on one box
OUT=gpurun_out/${OUTDIR:-r3zz2}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "msm" > $OUT/pytest_msm.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_msm.log
tail -3 $OUT/pytest_msm.log
run() { tag=$1; shift; timeout 400 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; }
AB="python tools/ab_kernels.py"
run fused1 $AB --parts msm --tag fused1
run plain1 GA_MSM_FUSE_MIN=1099511627776 $AB --parts msm --tag plain1
run fused2 $AB --parts msm --tag fused2
run plain2 GA_MSM_FUSE_MIN=1099511627776 $AB --parts msm --tag plain2
run g16fused $AB --parts g16 --tag g16fused
run g16plain GA_MSM_FUSE_MIN=1099511627776 $AB --parts g16 --tag g16plain
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/" + __import__("os").environ.get("OUTDIR", "r3zz2") + "/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "msm" in d:
        m = d["msm"]
        print(d["tag"], " ".join("%s msm %.3f acc %.3f reduce %.3f sort %.3f %s" % (g, m[g]["msm_ms"], m[g]["accumulate_ms"], m[g]["reduce_ms"], m[g]["sort_ms"], m[g]["sha"]) for g in ("g1", "g2")))
    if "g16" in d:
        print(d["tag"], d["g16"])
P
cat $OUT/failures.txt 2>/dev/null; true
