#!/bin/bash
# A/B of prebuilt library variants (gnark_amd/variants/lib_*.so): bench at 2^24 with the Groth16 leg
mkdir -p gpurun_out
export TMPDIR=/tmp
cp gnark_amd/libgnark_amd.so /tmp/lib_default.so
for v in gnark_amd/variants/lib_*.so; do
  cp $v gnark_amd/libgnark_amd.so
  timeout 600 python bench.py --no-cpu-baseline $VAR_ARGS > gpurun_out/var.log 2>&1
  python - "$v" <<'PY'
import json,sys
for line in open('gpurun_out/var.log'):
    if line.startswith('{'):
        d=json.loads(line); g=d.get('groth16',{})
        print(sys.argv[1], 'ms/step', d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['stages_ms'].items() if k in ('msm_accumulate','msm_reduce','msm_merge')}, 'g16', g.get('ms_per_proof'), 'g16 reduce', g.get('stages_ms',{}).get('msm_reduce',{}).get('total_ms'))
    elif 'rror' in line: print(line[:300])
PY
done
cp /tmp/lib_default.so gnark_amd/libgnark_amd.so
