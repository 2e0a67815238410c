# tools/shape1.sh <bench args ...>: one line of bench.py's main workload -- value, ms per step, kernel averages
timeout -k 5 ${T:-280} python bench.py --no-cpu-baseline --only-main --steps 3 --warmup 1 --no-verify "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '|', d['value'], d['ms_per_step'], {k:(round(v['ms_avg'],3), v['launches']) for k,v in d['kernels'].items()}, d['encoder_route'].get('text_entries'))"
