#!/bin/bash
# Round-3 batch ZZ: buckets per running-sum group of the window reduction (GA_MSM_GROUP: 32 shipped; 8 / 16 / 64), same box
OUT=gpurun_out/r3zz
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; }
AB="python tools/ab_kernels.py --parts msm"
run g32a $AB --tag g32a
run g16a GA_MSM_GROUP=16 $AB --tag g16a
run g8a  GA_MSM_GROUP=8 $AB --tag g8a
run g64a GA_MSM_GROUP=64 $AB --tag g64a
run g32b $AB --tag g32b
run g16b GA_MSM_GROUP=16 $AB --tag g16b
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3zz/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); m = d["msm"]
    except Exception as e:
        print(f, "unreadable", e); continue
    print(d["tag"], d["env"], " ".join("%s msm %.3f acc %.3f reduce %.3f sort %.3f %s" % (g, m[g]["msm_ms"], m[g]["accumulate_ms"], m[g]["reduce_ms"], m[g]["sort_ms"], m[g]["sha"]) for g in ("g1", "g2")))
P
cat $OUT/failures.txt 2>/dev/null
