#!/bin/bash
# round 6, batch I: timeline of ga_g16_prove_oneshot at 2^24 (GA_TRACE_PIN=1: uploader and MSM-wait events on one clock)
export TAG=r06_i
OUT=gpurun_out
GA_TRACE_PIN=1 python - > $OUT/r06_i_oneshot.txt 2> $OUT/r06_i_oneshot.err <<'PY'
import sys, time
sys.path.insert(0, ".")
from gnark_amd import synth
from gnark_amd.device import Context
ctx = Context(0)
inst = synth.make_instance(ctx, "bn254", 24, 0x5EED0005, want_dlogs=False)
for k in range(3):
    t0 = time.perf_counter()
    inst.prove_oneshot(ctx)
    print("round", k, round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
    sys.stderr.write("---- round %d done\n" % k)
PY
cat $OUT/r06_i_oneshot.txt; grep -v "^\[pin\]" $OUT/r06_i_oneshot.err | tail -22
