#!/bin/bash
mkdir -p gpurun_out
for a in "--mean 2048 --entries 2000000" "--mean 2000 --entries 2000000" "--mean 1024 --entries 4000000" "--mean 1000 --entries 4000000" "--mean 4096 --entries 1000000" "--mean 4000 --entries 1000000"; do
  python bench.py --no-cpu-baseline --only-main --no-walk-index --no-verify --steps 3 --warmup 1 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], json.dumps({k:v['ms_avg'] for k,v in d['kernels'].items()}))
"
done > gpurun_out/probe_shapes3.txt 2>&1
cat gpurun_out/probe_shapes3.txt
