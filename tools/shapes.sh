for a in "--mean 300 --entries 4000000" "--mean 2000 --entries 2000000" "--del-run-p 0.5 --sub-run-p 0.5" "--del-run-p 0.3 --sub-run-p 0.3"; do
  timeout -k 5 280 python bench.py --no-cpu-baseline --only-main --steps 3 --warmup 1 --no-verify $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], {k:(round(v['ms_avg'],3), v['launches']) for k,v in d['kernels'].items()}, d['encoder_route'].get('text_entries'))"
done
