for v in "" ml64k ml128k; do
  if [ -z "$v" ]; then unset GA_LIB_PATH; else export GA_LIB_PATH=$PWD/gnark_amd/variants/libgnark_amd_$v.so; fi
  python tools/bench_plonk_kernels.py 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant=$v plonk', d['ms_per_proof_kernels'], d['msm_ms'], {k:v for k,v in d['stages_ms'].items() if k.startswith('msm_')})"
  for ln in 20 24; do python bench.py --log-n $ln --no-cpu-baseline --no-check --groth16-proofs 0 --plonk-log-n 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   2^$ln', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages_ms'].items() if k in ('msm_reduce','msm_merge')})"; done
done
