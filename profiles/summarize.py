#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_*, written by profiles/tools/profile_all.sh) into committed summaries.

    python profiles/summarize.py <round-tag> [gpurun_out]

Writes, under profiles/:
  <tag>_kernel_stats.csv, <tag>_kernel_stats_dexta.csv, <tag>_kernel_stats_dexar.csv
        rocprofv3 --kernel-trace --stats summaries, verbatim
  <tag>_traffic.json (and traffic.json, which bench.py reads for roofline.traffic)
        HBM bytes per launch of every kernel from the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
        kernel trace only), corrected as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950
        FETCH_SIZE reports half the bytes of a wide 16 B/lane streaming read, so it is doubled
  <tag>_sq_counters.json
        SQ counters per launch of the dexqv kernels on a 200 k-entry batch (instructions per 5 KiB wave-step)
"""
import csv, glob, json, os, shutil, sys, collections

# device kernel -> the id bench.py times it under (dx_kernel_name)
BENCH_ID = {"k_qv_encode_fast": "k_qv_encode", "k_qv_encode": "k_qv_encode_text", "k_qv_hist": "k_qv_hist",
            "k_qv_compact": "k_qv_compact", "k_qv_decode": "k_qv_decode", "k_qv_decode_tags": "k_qv_decode_tags",
            "k_qv_decode_plain": "k_qv_decode_plain", "k_qv_decode_sub": "k_qv_decode_sub", "k_qv_decode_runs": "k_qv_decode_runs",
            "k_qv_sizes_fast": "k_qv_sizes", "k_qv_sizes": "k_qv_sizes",
            "k_qv_prescan_del": "k_qv_prescan", "k_qv_prescan_sub": "k_qv_prescan",
            "k_scan_tiles": "k_scan", "k_scan_sums": "k_scan", "k_scan_apply": "k_scan", "k_scan_apply_base": "k_scan",
            "k_qv_bounds": "k_scan", "k_tok_rooms": "k_scan", "k_sub_rooms": "k_scan",
            "k_pack2_encode": "k_pack2_encode", "k_pack2_decode": "k_pack2_decode"}


def kname(full):
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def counters(src, sub):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def traffic(src, suffix, steps):
    fe, wr = counters(src, "prof_fetch" + suffix), counters(src, "prof_write" + suffix)
    out, lps = {}, {}
    for name in sorted(set(fe) | set(wr)):
        if not name.startswith("k_"):
            continue
        f = fe.get(name, {}).get("FETCH_SIZE", [])
        w = wr.get(name, {}).get("WRITE_SIZE", [])
        # the x2 is for wide coalesced streams (16 B per lane, consecutive lanes consecutive bytes): every kernel here
        # but the lane-per-stream decoder, whose lanes each read their own line (raw request bytes kept for it)
        scattered = name in ("k_qv_decode", "k_qv_decode_plain", "k_qv_decode_sub", "k_qv_decode_runs")
        out[name] = {"fetch_bytes": sum(f) / len(f) * 1024 * (1 if scattered else 2) if f else None,
                     "fetch_bytes_raw_counter": sum(f) / len(f) * 1024 if f else None,
                     "write_bytes": sum(w) / len(w) * 1024 if w else None,
                     "launches_sampled": len(f) or len(w)}
        out[name]["hbm_bytes_per_launch"] = (out[name]["fetch_bytes"] or 0) + (out[name]["write_bytes"] or 0)
        lps[name] = max(1, round(out[name]["launches_sampled"] / steps))
    # per bench kernel id: bytes and launches of an id are the sums over its device kernels, so that bytes per launch
    # there and ms per launch in bench.py average over the same launches (since round 3 every main kernel has an id of
    # its own: the fast encoder "k_qv_encode", the text-reading one "k_qv_encode_text", each decode kernel its name)
    ids = collections.defaultdict(lambda: {"hbm_bytes_per_step": 0.0, "launches_per_step": 0})
    for name, v in out.items():
        b = BENCH_ID.get(name)
        if b:
            ids[b]["hbm_bytes_per_step"] += v["hbm_bytes_per_launch"] * lps[name]
            ids[b]["launches_per_step"] += lps[name]
    return out, lps, ids


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
    here = os.path.dirname(os.path.abspath(__file__))
    for suffix in ("", "_dexta", "_dexar"):
        ks = glob.glob(os.path.join(src, "prof_stats" + suffix, "*", "*_kernel_stats.csv"))
        if ks:
            shutil.copy(ks[0], os.path.join(here, f"{tag}_kernel_stats{suffix}.csv"))
    steps = 4                                            # PMC passes: warmup 1 + steps 3
    doc = {"tag": tag, "units": "bytes", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count on 16 B/lane streams); WRITE_SIZE KiB x 1024",
           "workloads": {}}
    k, lps, ids = traffic(src, "", steps)
    dq = {"entries": 1000000, "mean": 10000, "dist": "fixed", "kernels": dict(k), "launches_per_step": dict(lps)}
    for b, v in ids.items():                             # what bench.py looks up: bytes per launch of its kernel ids
        n = v["launches_per_step"]
        dq["kernels"].setdefault(b, {})
        dq["kernels"][b] = dict(dq["kernels"][b], hbm_bytes_per_launch=v["hbm_bytes_per_step"] / n, hbm_bytes_per_step=v["hbm_bytes_per_step"])
        dq["launches_per_step"][b] = n
    doc["workloads"]["dexqv"] = dq
    for w in ("dexta", "dexar"):
        k, lps, ids = traffic(src, "_" + w, steps)
        if k:
            doc["workloads"][w] = {"reads": 10000000, "mean": 10000, "kernels": k, "launches_per_step": lps}
    json.dump(doc, open(os.path.join(here, f"{tag}_traffic.json"), "w"), indent=1)
    json.dump(doc, open(os.path.join(here, "traffic.json"), "w"), indent=1)
    sq = {}
    for i in (1, 2):
        for name, c in counters(src, f"prof_sq{i}").items():
            if name.startswith("k_qv_") or name.startswith("k_pack2"):
                sq.setdefault(name, {}).update({kk: round(sum(v) / len(v)) for kk, v in c.items()})
    for name, c in sq.items():
        if "SQ_INSTS_VALU" in c:
            c["VALU_per_5KiB_wave_step"] = round(c["SQ_INSTS_VALU"] / 2e6, 1)     # 200 k entries x 10 steps
    json.dump({"tag": tag, "batch": "200000 x 10000", "per_launch": sq}, open(os.path.join(here, f"{tag}_sq_counters.json"), "w"), indent=1)
    print(json.dumps({w: {kk: round(vv["hbm_bytes_per_launch"] / 1e9, 3) for kk, vv in d["kernels"].items()} for w, d in doc["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
