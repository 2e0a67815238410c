for rep in 1 2; do
for sp in "" "55,45" "60,40" "65,35" "70,30"; do
  DEXGPU_ONEPASS_SPLIT=$sp python bench.py --no-cpu-baseline --only-main --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
print('split=[$sp]', d['value'], d['ms_per_step'], d.get('roundtrip_bit_exact'), {k:v['ms_per_step'] for k,v in pk.items() if v['ms_per_step']>0.5})"
done; done
