#!/bin/bash
# A/B of library builds for the 2-bit packers: tools/ab_pack2.sh <variant> ...  (see tools/ab.sh)
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/dextractor_amd/libdexgpu.so
    [ "$v" != main ] && lib=$PWD/tools/variants/libdexgpu_$v.so
    DEXGPU_LIB=$lib python bench.py --workload dexta --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', 'enc', d['encode_ms'], d['roofline']['frac'], 'dec', d['decode']['ms'], d['decode']['frac'], d['roundtrip_bit_exact'])"
  done
done
