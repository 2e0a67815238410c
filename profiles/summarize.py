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
            "k_qv_sizes_fast": "k_qv_sizes", "k_qv_sizes": "k_qv_sizes", "k_qv_sizes_hist": "k_qv_sizes",
            "k_walk_find": "k_qv_walk", "k_walk_pieces": "k_qv_walk", "k_walk_gather": "k_qv_walk", "k_walk_rooms": "k_qv_walk", "k_walk_index": "k_qv_walk",
            "k_qv_decode_sync": "k_qv_decode_sub",
            "k_qv_prescan_del": "k_qv_prescan", "k_qv_prescan_sub": "k_qv_prescan",
            "k_scan_tiles": "k_scan", "k_scan_sums": "k_scan", "k_scan_apply": "k_scan", "k_scan_apply_base": "k_scan",
            "k_qv_bounds": "k_scan", "k_sub_rooms": "k_scan",
            "k_pack2_encode": "k_pack2_encode", "k_pack2_decode": "k_pack2_decode"}


# how a kernel's lanes read (see traffic()): "stream" = 16 B a lane, consecutive lanes consecutive bytes; "lanes" = every lane its own
ACCESS = {"k_qv_hist": "stream", "k_qv_encode_fast": "stream", "k_qv_encode": "stream", "k_qv_compact": "stream", "k_qv_sizes_hist": "stream",
          "k_qv_sizes_fast": "stream", "k_qv_sizes": "stream", "k_qv_prescan_del": "stream", "k_qv_prescan_sub": "stream", "k_qv_density": "stream",
          "k_qv_bounds": "stream", "k_sub_rooms": "stream", "k_scan_tiles": "stream", "k_scan_sums": "stream", "k_scan_apply": "stream",
          "k_scan_apply_base": "stream", "k_ticket_units": "lanes", "k_qv_lossy_text": "stream",
          "k_pack2_encode": "stream", "k_pack2_decode": "stream", "k_synth_quiva": "stream", "k_synth_seq": "stream",
          "k_nl_count": "stream", "k_nl_fill": "stream", "k_qv_entries": "lanes", "k_seq_lines": "lanes", "k_seq_records": "lanes", "k_seq_extent": "lanes",
          "k_qv_decode": "lanes", "k_qv_decode_plain": "lanes", "k_qv_decode_sub": "lanes", "k_qv_decode_runs": "lanes", "k_qv_decode_sync": "lanes",
          "k_qv_decode_tags": "stream",
          "k_add_one": "lanes", "k_gather_headers": "lanes", "k_gather_lines": "lanes", "k_tok_rooms": "stream",
          "k_walk_find": "stream", "k_walk_pieces": "lanes", "k_walk_gather": "lanes", "k_walk_rooms": "lanes", "k_walk_index": "lanes",
          "k_qs_survey": "lanes", "k_qs_hist": "lanes", "k_qs_entries": "lanes", "k_qs_lenhist": "lanes", "k_qs_cut": "lanes",
          "k_qs_round_order": "lanes", "k_qs_sub_batch": "lanes", "k_qs_scatter_sizes": "lanes", "k_qs_gather_places": "lanes",
          "k_scan_state": "lanes"}


def kname(full):
    n = full.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    return n.split("<")[0]


def newest(paths):
    """gpurun merges a call's files into gpurun_out/ and never removes older ones: of several runs' files, the last one's"""
    paths = sorted(paths, key=os.path.getmtime)
    return paths[-1:] if paths else []


def counters(src, sub):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            per[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def bench_launches(src, suffix):
    """launches per step of every bench kernel id in the profiled run itself (its JSON line: launches over the timed steps)"""
    try:
        line = [ln for ln in open(os.path.join(src, "prof_fetch" + suffix + ".json")) if ln.startswith("{")][0]
        d = json.loads(line)
        return {k: max(1, round(v["launches"] / d["steps"])) for k, v in d.get("kernels", {}).items()}
    except Exception:
        return {}


def traffic(src, suffix):
    fe, wr = counters(src, "prof_fetch" + suffix), counters(src, "prof_write" + suffix)
    out = {}
    for name in sorted(set(fe) | set(wr)):
        if not name.startswith("k_"):
            continue
        f = fe.get(name, {}).get("FETCH_SIZE", [])
        w = wr.get(name, {}).get("WRITE_SIZE", [])
        # The guide's x2 is for wide coalesced streams (16 B per lane, consecutive lanes consecutive bytes).  Every kernel is
        # classified by how its lanes read, in ACCESS below; a kernel that is not listed fails the summary (a new kernel must be
        # looked at, not inherit a correction): "stream" x2; "lanes" (every lane its own line or piece: the lane-per-line decoders,
        # the walk's lanes, the window loads of the wave-per-line decoders, which are 4 B a lane) raw request bytes.
        if name not in ACCESS:
            raise SystemExit(f"profiles/summarize.py: kernel {name} has no entry in ACCESS (how do its lanes read?)")
        scattered = ACCESS[name] == "lanes"
        out[name] = {"fetch_bytes": sum(f) / len(f) * 1024 * (1 if scattered else 2) if f else None,
                     "fetch_bytes_raw_counter": sum(f) / len(f) * 1024 if f else None,
                     "fetch_correction": 1 if scattered else 2, "access": ACCESS[name],
                     "write_bytes": sum(w) / len(w) * 1024 if w else None,
                     "rows_sampled": len(f) or len(w)}
        out[name]["hbm_bytes_per_launch"] = (out[name]["fetch_bytes"] or 0) + (out[name]["write_bytes"] or 0)
    # per bench kernel id (dx_kernel_name: what bench.py times): bytes per launch = the average over the sampled launches
    # of the id's device kernels; launches per step from the profiled run's own JSON line
    lps = bench_launches(src, suffix)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for name, v in out.items():
        b = BENCH_ID.get(name)
        if b:
            acc[b][0] += v["hbm_bytes_per_launch"] * v["rows_sampled"]
            acc[b][1] += v["rows_sampled"]
    ids = {}
    for b, (tot, rows) in acc.items():
        if rows:                                           # (kernels outside the timed steps -- the decoders, the packers -- run once)
            n = lps.get(b, 1)
            ids[b] = {"hbm_bytes_per_launch": tot / rows, "launches_per_step": n, "hbm_bytes_per_step": tot / rows * n}
    return out, ids


def main():
    tag = sys.argv[1]
    src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
    here = os.path.dirname(os.path.abspath(__file__))
    for suffix in ("", "_dexta", "_dexar"):
        ks = newest(glob.glob(os.path.join(src, "prof_stats" + suffix, "*", "*_kernel_stats.csv")))
        if ks:
            shutil.copy(ks[0], os.path.join(here, f"{tag}_kernel_stats{suffix}.csv"))
    doc = {"tag": tag, "units": "bytes", "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count on 16 B/lane streams); WRITE_SIZE KiB x 1024",
           "workloads": {}}
    k, ids = traffic(src, "")
    doc["workloads"]["dexqv"] = {"entries": 1000000, "mean": 10000, "dist": "fixed", "kernels": dict(k), "bench_ids": ids}
    for w in ("dexta", "dexar"):
        k, ids = traffic(src, "_" + w)
        if k:
            doc["workloads"][w] = {"reads": 10000000, "mean": 10000, "kernels": k, "bench_ids": ids}
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
    print(json.dumps({w: {kk: round(vv["hbm_bytes_per_step"] / 1e9, 3) for kk, vv in d["bench_ids"].items()} for w, d in doc["workloads"].items()}, indent=1))


if __name__ == "__main__":
    main()
