#!/bin/bash
mkdir -p gpurun_out
for a in "--mean 10240" "--mean 10000" "--mean 9216" "--mean 9000"; do
  python bench.py --no-cpu-baseline --only-main --no-walk-index --no-verify --steps 4 --warmup 1 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], json.dumps({k:round(v['ms_avg'],3) for k,v in d['kernels'].items()}))
"
done > gpurun_out/probe_shapes4.txt 2>&1
cat gpurun_out/probe_shapes4.txt
