#!/bin/bash
# tools/dev/fast_ab.sh <variant> ...: the headline step and the 2 M x 2 kb shape under each variant library (tools/mkvariant.sh)
mkdir -p gpurun_out
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib="tools/variants/libdexgpu_$v.so"
  for shape in "1000000 10000 8" "2000000 2000 4" "4000000 600 4"; do
    set -- $shape
    DEXGPU_LIB=$lib DEXGPU_NO_SHORT=1 python bench.py --entries $1 --mean $2 --steps $3 --warmup 2 --only-main --no-cpu-baseline --no-walk-index --no-verify > gpurun_out/fab.json 2> gpurun_out/fab.err || { tail -3 gpurun_out/fab.err; continue; }
    python - "$v" $1 $2 <<P
import json,sys
d=json.load(open("gpurun_out/fab.json"))
print(sys.argv[1], sys.argv[2], "x", sys.argv[3], d["value"], "GB/s", d["ms_per_step"], "ms", d["roundtrip_bit_exact"], {k: round(v["ms_avg"],2) for k,v in d["kernels"].items()})
P
  done
done
