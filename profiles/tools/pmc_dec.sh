#!/bin/bash
# usage (repo root, GPU box): bash profiles/tools/pmc_dec.sh "<counter group>" ...   (counters of the decode kernels)
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmcd_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcd_$i -o p -f csv -- python3 $R/bench.py --only-main --no-check --no-cpu-baseline --entries 200000 --steps 1 --warmup 1 > /tmp/pmcd_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmcd_$i/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0]
        if 'decode' in n:
            per[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in sorted(per.items()):
    print(n, {k: round(sum(v)/len(v)) for k,v in sorted(c.items())})
PY
done
