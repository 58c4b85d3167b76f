for v in "" chunk256 chunk128; do
  if [ -z "$v" ]; then unset GA_LIB_PATH; else export GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_$v.so; fi
  python bench.py --no-cpu-baseline --no-check --groth16-proofs 3 --plonk-log-n 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['groth16']; print('variant=$v', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages_ms'].items() if k in ('msm_merge','msm_reduce','msm_accumulate')}, g['ms_per_proof'], {k:v['total_ms'] for k,v in g['stages_ms'].items() if k in ('msm_merge','msm_reduce')})"
done
