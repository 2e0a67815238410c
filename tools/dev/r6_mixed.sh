#!/bin/bash
# tools/dev/r6_mixed.sh: the mixed-length workload and its parts, with and without the mixed route
O=gpurun_out; mkdir -p $O
B="python bench.py --only-main --no-cpu-baseline --no-walk-index --steps 3 --warmup 1"
run() { tag=$1; shift; env "$@" > $O/mx_$tag.json 2> $O/mx_$tag.err; python - $tag <<P
import json,sys
try:
    d=json.loads([l for l in open("gpurun_out/mx_%s.json"%sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], d["value"], "GB/s", d["ms_per_step"], "ms", d["roundtrip_bit_exact"], d["encoder_route"].get("direct"), {k: round(v["ms_avg"],3) for k,v in d["kernels"].items()})
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/mx_%s.err"%sys.argv[1]).read()[-400:])
P
}
run mixed X=1 $B --dist mixed --entries 2000000
run mixed_off DEXGPU_TEST=no_mixed $B --dist mixed --entries 2000000
run short_part X=1 $B --dist short_u --entries 1800000
run long_part X=1 $B --dist lognormal --entries 200000
run e2000 X=1 $B --entries 2000000 --mean 2000
run e300 X=1 $B --entries 4000000 --mean 300
