#!/bin/bash
# Round-3 batch H: stream priorities for the lanes (GA_LANE_PRIO unset / 1 = partner lanes high / 2 = witness lanes high), same box
OUT=gpurun_out/r3h
mkdir -p $OUT
export TMPDIR=/tmp
run() { tag=$1; shift; timeout 500 env "$@" > $OUT/ab_$tag.json 2> $OUT/ab_$tag.err || echo "FAILED $tag rc=$?" >> $OUT/failures.txt; tail -c 250 $OUT/ab_$tag.err; }
AB="python tools/ab_kernels.py"
run p0   $AB --parts g16 --tag p0 --proofs 8
run p1   GA_LANE_PRIO=1 $AB --parts g16 --tag p1 --proofs 8
run p2   GA_LANE_PRIO=2 $AB --parts g16 --tag p2 --proofs 8
run p0b  $AB --parts g16 --tag p0b --proofs 8
run p1b  GA_LANE_PRIO=1 $AB --parts g16 --tag p1b --proofs 8
python - <<'P' > $OUT/ab_summary.txt 2>&1
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h/ab_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(d["tag"], d["env"], d.get("g16"))
P
cat $OUT/ab_summary.txt
