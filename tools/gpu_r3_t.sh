#!/bin/bash
OUT=gpurun_out/r3t
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/exp/one_shot_breakdown.py 24 > $OUT/one_shot_breakdown.jsonl 2> $OUT/breakdown.err; cat $OUT/one_shot_breakdown.jsonl; tail -2 $OUT/breakdown.err
