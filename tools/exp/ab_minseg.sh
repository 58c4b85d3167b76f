for sg in 256 512 1024 2048 4096; do
  GA_MSM_MIN_SEG=$sg python bench.py --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['groth16']; print('min_seg=$sg', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages_ms'].items() if k in ('msm_tasks','msm_accumulate','msm_merge','msm_reduce')}, 'g16', g['ms_per_proof'], {k:v['total_ms'] for k,v in g['stages_ms'].items() if k in ('msm_accumulate','msm_merge','msm_tasks')})"
done
