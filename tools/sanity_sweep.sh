#!/bin/bash
# usage (repo root, GPU box): bash tools/sanity_sweep.sh -- short bench runs over workload shapes; every line must say True where it checks
# (round trip through the lane-per-line decoders, through the indexed ones with the encoder's index and with the host walk's, and
#  the device walk of the stream: its index the encoder's, the decode from it bit-exact)
fail=0
while read -r a; do
  python bench.py --no-cpu-baseline --only-main --steps 2 --warmup 1 $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
g=lambda k,f: (d.get(k) or {}).get(f)
checks = [d.get('roundtrip_bit_exact'), g('decode_indexed','bit_exact'), g('decode_walk_indexed','bit_exact'), g('device_walk','index_identical_to_the_encoders'), g('device_walk','decode_bit_exact')]
print('$a |', d['value'], d['ms_per_step'], checks[0], '| decode', g('decode','ms'), '| indexed', g('decode_indexed','ms'), checks[1], '| walk', g('decode_walk_indexed','ms'), checks[2],
      '| device walk', g('device_walk','kernel_ms'), '+', g('device_walk','decode_ms_from_this_index'), checks[3], checks[4], 'lines without', g('device_walk','run_lines_without_groups'), g('device_walk','plain_lines_without_words'),
      '| OK' if all(c is True for c in checks) else '| FAILED (a check is not True)')
sys.exit(0 if all(c is True for c in checks) else 1)" || fail=1
done <<'L'
--dist lognormal
--lossy
--del-run-p 0.99 --sub-run-p 0.99
--mean 300 --entries 4000000
--mean 60000 --entries 150000
L
exit $fail
