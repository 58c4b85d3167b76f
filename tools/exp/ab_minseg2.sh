for ln in 18 20 22; do for sg in 256 1024; do
  GA_MSM_MIN_SEG=$sg python bench.py --log-n $ln --no-cpu-baseline --no-check --groth16-proofs 1 --no-pipelined --plonk-log-n 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); g=d['groth16']; print('2^$ln min_seg=$sg', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages_ms'].items() if k in ('msm_tasks','msm_accumulate','msm_merge','msm_reduce')}, 'g16', g['ms_per_proof'])"
done; done
for sg in 256 1024; do GA_MSM_MIN_SEG=$sg python tools/exp/hot_bucket.py 24 | grep boolean; GA_MSM_MIN_SEG=$sg python tools/bench_plonk_kernels.py 2>/dev/null | head -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('plonk min_seg=$sg', d['ms_per_proof_kernels'], d['msm_ms'])"; done
