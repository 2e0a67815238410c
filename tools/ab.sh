#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh <bench args> -- <variant> ...   (variant "main" = dextractor_amd/libdexgpu.so,
# otherwise tools/variants/libdexgpu_<variant>.so); every variant twice, interleaved; one line per run
args=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for rep in 1 2; do
  for v in "$@"; do
    lib=$PWD/dextractor_amd/libdexgpu.so
    [ "$v" != main ] && lib=$PWD/tools/variants/libdexgpu_$v.so
    DEXGPU_LIB=$lib python bench.py --no-cpu-baseline --only-main "${args[@]}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
pk=d['roofline']['per_kernel']
print('$v', d['value'], d['ms_per_step'], d.get('roundtrip_bit_exact'), {k:v['ms_per_step'] for k,v in pk.items() if v['ms_per_step']>0.5}, (d.get('decode') or {}).get('ms'), (d.get('decode_indexed') or {}).get('ms'), (d.get('decode_indexed') or {}).get('ms_by_kernel'))"
  done
done
