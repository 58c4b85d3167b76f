for ln in 20 22 24; do
for c in 0 17 18 19 20 21 22; do
  GA_TABLE_C=$c python bench.py --log-n $ln --no-cpu-baseline --no-check --groth16-proofs 0 --plonk-log-n 0 --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('2^$ln c=$c ->', d['config']['window_bits'], d['config']['windows'], 'ms', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['stages_ms'].items()})"
done; done
