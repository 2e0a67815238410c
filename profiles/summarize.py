#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_{stats,fetch,write}) into committed summaries.

    python profiles/summarize.py <round-tag> [gpurun_out] [--entries N --mean M --dist D]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
profiles/<tag>_traffic.json (PMC passes: FETCH_SIZE / WRITE_SIZE per launch, corrected as
MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE reports half the
bytes of a wide 16 B/lane streaming read, so it is doubled) and refreshes profiles/traffic.json
(read by bench.py for roofline.traffic).
"""
import csv, glob, json, os, shutil, sys, collections

def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "gpurun_out"
    opt = {"entries": 1000000, "mean": 10000, "dist": "fixed", "passes": 3}   # passes: warmup + steps of the PMC runs
    for i, a in enumerate(sys.argv):
        if a.startswith("--") and a[2:] in opt:
            opt[a[2:]] = type(opt[a[2:]])(sys.argv[i + 1])
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(src, "prof_stats", "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(here, f"{tag}_kernel_stats.csv"))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for which in ("fetch", "write"):
        for f in glob.glob(os.path.join(src, f"prof_{which}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0]
                per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"tag": tag, **opt, "units": "bytes per launch", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count on 16 B/lane streams); WRITE_SIZE KiB x 1024",
           "kernels": {}}
    for name, c in sorted(per.items()):
        if not name.startswith("k_"):
            continue
        fe = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) * 1024 * 2 if c.get("FETCH_SIZE") else None
        wr = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) * 1024 if c.get("WRITE_SIZE") else None
        out["kernels"][name] = {"fetch_bytes": fe, "write_bytes": wr,
                                "hbm_bytes_per_launch": (fe or 0) + (wr or 0),
                                "launches_sampled": len(c.get("FETCH_SIZE", []))}
    out["launches_per_step"] = {k: max(1, round(v["launches_sampled"] / opt["passes"])) for k, v in out["kernels"].items()}
    json.dump(out, open(os.path.join(here, f"{tag}_traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(here, "traffic.json"), "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))

if __name__ == "__main__":
    main()
