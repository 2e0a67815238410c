#!/bin/bash
mkdir -p gpurun_out
for a in "--del-run-p 0.99 --sub-run-p 0.99" "--del-run-p 0.97 --sub-run-p 0.97" "--del-run-p 0.95 --sub-run-p 0.95" "" "--del-run-p 0.999 --sub-run-p 0.999"; do
  python bench.py --no-cpu-baseline --only-main --no-walk-index --steps 5 --warmup 2 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], d['roundtrip_bit_exact'], json.dumps({k:v['ms_avg'] for k,v in d['kernels'].items()}))
"
done > gpurun_out/probe_shapes2.txt 2>&1
cat gpurun_out/probe_shapes2.txt
