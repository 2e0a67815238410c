#!/bin/bash
# per-kernel times of the off-headline shapes, and the tools' resident set on small inputs
mkdir -p gpurun_out
for a in "--del-run-p 0.99 --sub-run-p 0.99" "--del-run-p 0.95 --sub-run-p 0.95" "--del-run-p 0.5 --sub-run-p 0.5" "--mean 2000 --entries 2000000"; do
  python bench.py --no-cpu-baseline --only-main --no-walk-index --steps 3 --warmup 1 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$a |', d['value'], d['ms_per_step'], json.dumps(d['kernels']), json.dumps(d.get('encoder_route')))
"
done > gpurun_out/probe_shapes.txt 2>&1
python tools/dev/r6_rss.py > gpurun_out/probe_rss.txt 2>&1
cat gpurun_out/probe_shapes.txt gpurun_out/probe_rss.txt
