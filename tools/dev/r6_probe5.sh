#!/bin/bash
mkdir -p gpurun_out
for a in "--del-run-p 0.5 --sub-run-p 0.5" "--del-run-p 0.3 --sub-run-p 0.3" "--del-run-p 0.7 --sub-run-p 0.7" "--del-run-p 0.6 --sub-run-p 0.9" ""; do
  python bench.py --no-cpu-baseline --only-main --no-walk-index --steps 3 --warmup 1 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], d['roundtrip_bit_exact'], json.dumps({k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}), (d.get('encoder_route') or {}).get('text_entries'))
"
done > gpurun_out/probe_shapes5.txt 2>&1
cat gpurun_out/probe_shapes5.txt
