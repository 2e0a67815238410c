for rep in 1 2; do
for f in 0 2; do
  for dist in fixed lognormal; do
  DEXGPU_FOLLOW=$f python bench.py --no-cpu-baseline --only-main --steps 8 --warmup 2 --dist $dist 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
print('follow=$f $dist', d['value'], d['ms_per_step'], d.get('roundtrip_bit_exact'), {k:v['ms_per_step'] for k,v in pk.items() if v['ms_per_step']>0.5}, d.get('encoder_route',{}).get('chain_waits'), d.get('encoder_route',{}).get('groups'))"
done; done; done
