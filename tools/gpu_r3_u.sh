#!/bin/bash
OUT=gpurun_out/r3u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x -k "two_keys or two_callers or soak" > $OUT/t.log 2>&1; tail -4 $OUT/t.log
