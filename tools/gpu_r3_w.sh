#!/bin/bash
# Round-3 batch W: the final bench line once more (with the PCIe-inclusive MSM figure)
OUT=gpurun_out/r3w
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/r03_w_bench_bn254_2p24.json 2> $OUT/bench.err; tail -c 200 $OUT/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r3w/r03_w_bench_bn254_2p24.json").read().strip().splitlines()[-1])
print(d["value"], d["value_checked"], d["ms_per_step"], d["msm_with_scalar_h2d"], d["plain_msm_no_tables"], d["groth16"]["ms_per_proof"], d["groth16"]["pipelined"]["ms_per_proof"])
P
