#!/bin/bash
# usage (from the repo root, on the GPU box): bash profiles/tools/pmc.sh "<counter group 1>" "<counter group 2>" ...
# collects SQ counters for a 200k-entry bench run
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o p -f csv -- python3 $R/bench.py --only-main --no-check --no-cpu-baseline --no-verify --entries 200000 --steps 2 --warmup 1 > /tmp/pmc_$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_$i/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name'].split('(')[0]
        if n.startswith('k_qv_') or n.startswith('k_pack2'):
            per[n][r['Counter_Name']].append(float(r['Counter_Value']))
for n,c in sorted(per.items()):
    print(n, {k: round(sum(v)/len(v)) for k,v in sorted(c.items())})
PY
done
