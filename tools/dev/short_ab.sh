#!/bin/bash
# tools/dev/short_ab.sh <variant> ...  -- the short-entry workload (4 M x 300) under each variant library (tools/mkvariant.sh), kernel times
mkdir -p gpurun_out
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib="tools/variants/libdexgpu_$v.so"
  DEXGPU_LIB=$lib python bench.py --entries 4000000 --mean 300 --steps 4 --warmup 1 --only-main --no-cpu-baseline --no-walk-index --no-check --no-verify > gpurun_out/short_$v.json 2> gpurun_out/short_$v.err || { tail -3 gpurun_out/short_$v.err; continue; }
  python - "$v" <<P
import json,sys
d=json.load(open("gpurun_out/short_%s.json"%sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"], {k: round(v["ms_avg"],3) for k,v in d["kernels"].items()})
P
done
