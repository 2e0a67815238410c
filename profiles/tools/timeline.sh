#!/bin/bash
# usage (repo root, GPU box): bash profiles/tools/timeline.sh [extra bench flags] -- kernel trace of a short bench run,
# reduced to one line per launch (name, queue, start, end, duration in us) in gpurun_out/timeline.txt
R=$PWD
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/prof_tl
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -f csv -- python3 $R/bench.py --only-main --no-cpu-baseline --steps 3 --warmup 1 "$@" > $R/gpurun_out/prof_tl.json 2> $R/gpurun_out/prof_tl.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open("$R/gpurun_out/timeline.txt", "w") as o:
    for r in rows:
        o.write("%-40s q%-4s %12.1f %12.1f %10.1f\n" % (r["Kernel_Name"][:40], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $R/gpurun_out/prof_tl
grep -c . $R/gpurun_out/timeline.txt
