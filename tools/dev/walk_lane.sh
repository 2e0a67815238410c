#!/bin/bash
# tools/dev/walk_lane.sh: pieces a lane (DEXGPU_TEST=walk_per_lane=<n>) against the walk's time, fixed and lognormal lengths
for dist in fixed lognormal; do
  for pl in 2 4 8; do
    DEXGPU_TEST=walk_per_lane=$pl,walk_debug python bench.py --dist $dist --no-cpu-baseline --only-main --steps 2 --warmup 1 > gpurun_out/wl.json 2> gpurun_out/wl.err
    python - $dist $pl <<P
import json,sys,re
d=json.loads([l for l in open("gpurun_out/wl.json") if l.startswith("{")][0]); w=d["device_walk"]
e=open("gpurun_out/wl.err").read()
m=re.findall(r"\[walk\] (k_walk_find [^\n]*|\d+ pieces[^\n]*)", e)
print(sys.argv[1], "per lane", sys.argv[2], "kernels", w["kernel_ms"], "wall", w["wall_ms"], "decode", w["decode_ms_from_this_index"], "sum", w["walk_and_decode_ms"], w["decode_bit_exact"], w["index_identical_to_the_encoders"], "|", " | ".join(m[-2:]))
P
  done
done
