#!/bin/bash
# Round-3 batch Q: the two robustness tests (soak, device memory down to the reserve) + the repro script, after device_malloc
OUT=gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/exp/oom_repro.py > $OUT/oom_repro_after.log 2>&1; tail -8 $OUT/oom_repro_after.log | cut -c1-300
timeout 900 python -m pytest tests -q -m gpu -x -k "soak or out_of_device or error_behaviour or 2_26 or proving_key_file" > $OUT/robust.log 2>&1; tail -5 $OUT/robust.log
