#!/bin/bash
# tools/dev/short_cross.sh: where the lane-per-entry kernels stop paying -- GB/s of the dexqv step by mean entry length, with and without them
for m in 600 1000 1200; do
  for off in 0 1; do
    DEXGPU_TEST=no_short=$off python bench.py --entries 2000000 --mean $m --steps 3 --warmup 1 --only-main --no-cpu-baseline --no-walk-index --no-verify > gpurun_out/cross.json 2> gpurun_out/cross.err || { tail -3 gpurun_out/cross.err; continue; }
    python - $m $off <<P
import json,sys
d=json.load(open("gpurun_out/cross.json"))
print("mean", sys.argv[1], "no_short" if sys.argv[2]=="1" else "short   ", d["value"], "GB/s", d["ms_per_step"], "ms", {k: round(v["ms_avg"],2) for k,v in d["kernels"].items()})
P
  done
done
