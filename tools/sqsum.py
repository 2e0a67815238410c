"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, counter sums per dispatch (millions) and the kernel's duration.
usage: python tools/sqsum.py <dir> ... [--kernel substr]"""
import csv, glob, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
want = "hist"
for i, a in enumerate(sys.argv):
    if a == "--kernel": want = sys.argv[i + 1]
dirs = [d for d in dirs if d != want]
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if want not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            n = len(disp[k])
            print(k, "dispatches", n, {c: round(x / n / 1e6, 2) for c, x in sorted(v.items())})
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        t = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if want in k: t[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        for k, v in t.items(): print(k, "ms", [round(x, 3) for x in v])
