for rep in 1 2; do
for f in 0 1; do
  for dist in fixed lognormal; do
  DEXGPU_HYBRID=$f python bench.py --no-cpu-baseline --only-main --no-walk-index --steps 8 --warmup 2 --dist $dist 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
print('hybrid=$f $dist', d['value'], d['ms_per_step'], d.get('roundtrip_bit_exact'), {k:v['ms_per_step'] for k,v in pk.items() if v['ms_per_step']>0.5})"
done; done; done
