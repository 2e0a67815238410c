#!/bin/bash
# tools/dev/short_n.sh: how many entries a batch of short ones needs before the lane-per-entry kernels pay (ms per step with / without them)
for cfg in "16384 300" "65536 300" "262144 300" "65536 1000" "262144 1000"; do
  set -- $cfg
  for ns in 0 1; do
    DEXGPU_TEST=no_short=$ns python bench.py --entries $1 --mean $2 --steps 20 --warmup 3 --only-main --no-cpu-baseline --no-walk-index --no-verify > gpurun_out/sn.json 2>/dev/null
    python - $1 $2 $ns <<P
import json,sys
d=json.load(open("gpurun_out/sn.json"))
print(sys.argv[1], "x", sys.argv[2], "no_short" if sys.argv[3]=="1" else "short   ", d["ms_per_step"], "ms", d["value"], "GB/s", d["encoder_route"]["direct"], {k: round(v["ms_avg"],3) for k,v in d["kernels"].items()})
P
  done
done
