#!/bin/bash
# tools/dev/r6_hist_ab.sh <variant> ...: k_qv_hist alone (tools/microbench/hist_time.py, 1 M x 10 kb, 6 launches) under each variant library, twice,
# interleaved with the main build; then one checked step of the bench per variant (round trip bit-exact?)
O=gpurun_out; mkdir -p $O
for rep in 1 2; do
  for v in main "$@"; do
    lib=""; [ "$v" != main ] && lib="$PWD/tools/variants/libdexgpu_$v.so"
    DEXGPU_LIB=$lib python tools/microbench/hist_time.py 1000000 6 2>$O/hab.err | tail -1 | sed "s/^/$v /"
  done
done
for v in "$@"; do
  DEXGPU_LIB=$PWD/tools/variants/libdexgpu_$v.so python bench.py --only-main --no-cpu-baseline --no-walk-index --steps 4 --warmup 2 > $O/hab_$v.json 2> $O/hab_$v.err
  python - $v <<P
import json,sys
try:
    d=json.loads([l for l in open("gpurun_out/hab_%s.json"%sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], d["ms_per_step"], "roundtrip", d.get("roundtrip_bit_exact"), {k: round(v["ms_avg"],3) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
P
done
