#!/usr/bin/env python3
"""profiles/check.py -- is the committed evidence set a function of the tree?  (run by tests/test_profiles.py, no GPU needed)

  1. profiles/MANIFEST.json names the set (tag, commit, files) and the SHA-256 of the device sources (dextractor_amd/csrc/*.hip,
     *.hpp, dx_layout.h) it was measured on: a kernel edited after the set was made fails the check until the set is made again
     (gpurun -- bash tools/evidence.sh; python tools/evidence_collect.py <tag>).
  2. Every kernel's rocprofv3 average (<tag>_kernel_stats*.csv, `--kernel-trace --stats`) must agree
       - within 3 % with the per-kernel times the PROFILED run itself reported (<tag>_profiled_bench*.json: same launches,
         HIP events against the profiler's timestamps; 8 % for k_qv_compact, whose launches share the GPU with the encoder
         of the next group on another stream: the events bracket what the stream saw, the profiler the kernel alone), and
       - within 7 % with the un-profiled bench line of the set (<tag>_bench*.json; profiled runs clock a little lower;
         20 % for k_qv_decode_plain, a lane-per-line kernel whose time moves that much from run to run -- and from box to box: the
         two runs are two leases -- with where the chunks of the verification land in memory),
     kernel by kernel: the `frac` figures of DESIGN.md section 5 are (bytes in the bench line) / (these times).
  3. The HBM traffic DESIGN.md quotes per kernel stands in one table tied to the set's traffic file (design_traffic)."""
import csv, glob, hashlib, json, os, sys

BENCH_ID = {"k_qv_encode_fast": "k_qv_encode", "k_qv_encode": "k_qv_encode_text", "k_qv_hist": "k_qv_hist", "k_qv_compact": "k_qv_compact"}


def device_sources_hash(root):
    h = hashlib.sha256()
    base = os.path.join(root, "dextractor_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.hpp")) + [os.path.join(base, "dx_layout.h")]):
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(open(p, "rb").read())
    return h.hexdigest()


def stats_avg_ms(path):
    out = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"].split("(")[0].strip()
        if name.startswith("void "):
            name = name[5:]
        calls, total, longest = int(r["Calls"]), float(r["TotalDurationNs"]), float(r.get("MaxNs") or 0.0)
        for key in {name, name.split("<")[0]}:                # every instance of a template on its own, and all of them together
            a = out.setdefault(key, [0, 0.0, 0.0])
            a[0] += calls; a[1] += total; a[2] = max(a[2], longest)
    return {k: (v[1] / v[0] / 1e6, v[0], v[1] / 1e6, v[2] / 1e6) for k, v in out.items()}          # avg ms, calls, total ms, longest call ms


def rel(a, b):
    return abs(a - b) / max(a, b)


def check(root):
    prof = os.path.join(root, "profiles")
    mpath = os.path.join(prof, "MANIFEST.json")
    if not os.path.isfile(mpath):
        return ["profiles/MANIFEST.json is missing"]
    m = json.load(open(mpath))
    problems = []
    if m["device_sources_sha256"] != device_sources_hash(root):
        problems.append(f"the device sources have changed since the evidence set {m['tag']} was made (commit {m['head'][:10]}): "
                        "run tools/evidence.sh on the GPU box and tools/evidence_collect.py again")
    f = lambda k: os.path.join(prof, m["files"][k]) if k in m["files"] else None
    load = lambda k: json.load(open(f(k))) if f(k) and os.path.isfile(f(k)) else None

    def compare(what, rocprof_ms, bench_ms, tol):
        if rocprof_ms is None or bench_ms is None or bench_ms < 0.05:
            return
        if rel(rocprof_ms, bench_ms) > tol:
            problems.append(f"{what}: rocprofv3 average {rocprof_ms:.3f} ms, bench line {bench_ms:.3f} ms (> {int(tol * 100)} % apart)")

    # dexqv: encode-path kernels, per launch
    if f("kernel_stats.csv"):
        st = stats_avg_ms(f("kernel_stats.csv"))
        for which, tol in (("profiled_bench.json", 0.03), ("bench.json", 0.07)):
            d = load(which)
            if d is None:
                problems.append(f"{which} of the set is missing"); continue
            for dev, bid in BENCH_ID.items():
                if dev in st and bid in d.get("kernels", {}) and d["kernels"][bid]["launches"]:
                    # (the stats run also holds the one untimed step that writes the group index: the same kernels -- but for the
                    #  encoder, whose index-writing instance <true> does more than the timed <false>: that one alone is compared)
                    inst = dev + "<false>" if dev == "k_qv_encode_fast" and dev + "<false>" in st else dev
                    avg, calls, total, longest = st[inst]
                    # (the profiler saw the warm-up launches too -- the process's first launch of a kernel is its slowest by a tenth --, the bench
                    #  line averages the timed steps: with more calls than timed launches the longest call is left out of the profiler's average)
                    if calls > d["kernels"][bid]["launches"] and calls > 1:
                        avg = (total - longest) / (calls - 1)
                    compare(f"{dev} [{which}]", avg, d["kernels"][bid]["ms_avg"], tol)
            # the decoders: totals over the passes of the verification (launch counts differ from pass to pass)
            if which == "profiled_bench.json":
                tot = {}
                for key in ("decode", "decode_indexed", "decode_walk_indexed", "device_walk"):
                    for part in ("ms_by_kernel", "ms_by_kernel_without_the_groups"):      # (the walk's index is decoded with and without its words)
                        for k, v in ((d.get(key) or {}).get(part) or {}).items():
                            tot[k] = tot.get(k, 0.0) + v
                for k, v in tot.items():
                    if k in st:
                        # (the library times k_qv_decode_sync, the walk index's plain-line decoder, under k_qv_decode_sub's id)
                        also = st["k_qv_decode_sync"][2] if k == "k_qv_decode_sub" and "k_qv_decode_sync" in st else 0.0
                        compare(f"{k} (all passes) [{which}]", st[k][2] + also, v, 0.05)
            else:
                for key, kernels in (("decode_indexed", ("k_qv_decode_sub", "k_qv_decode_runs")), ("decode", ("k_qv_decode_plain",))):
                    for k in kernels:
                        v = ((d.get(key) or {}).get("ms_by_kernel") or {}).get(k)
                        if v and k in st:
                            compare(f"{k} [{which}: {key}]", st[k][0], v, 0.20 if k == "k_qv_decode_plain" else 0.07)
    for w in ("dexta", "dexar"):
        if f(f"kernel_stats_{w}.csv"):
            st = stats_avg_ms(f(f"kernel_stats_{w}.csv"))
            for which, tol in ((f"profiled_bench_{w}.json", 0.03), (f"bench_{w}.json", 0.07)):
                d = load(which)
                if d is None:
                    problems.append(f"{which} of the set is missing"); continue
                compare(f"k_pack2_encode [{which}]", st.get("k_pack2_encode", (None,))[0], d.get("encode_ms"), tol)
                compare(f"k_pack2_decode [{which}]", st.get("k_pack2_decode", (None,))[0], (d.get("decode") or {}).get("ms"), tol)
    problems += design_traffic(root, m)
    return problems


def design_traffic(root, m):
    """3. DESIGN.md quotes HBM traffic per kernel in ONE table, headed by a line `<!-- traffic-table: profiles/<file> -->`; every row
    `| kernel | GB per launch |` must be within 5 % of that file's figure (workloads.dexqv.kernels.<kernel>.hbm_bytes_per_launch),
    and the file must be the set's own: a traffic figure in the text with no file behind it, or an old file's, fails here."""
    import re
    out = []
    try:
        text = open(os.path.join(root, "DESIGN.md")).read()
    except OSError:
        return ["DESIGN.md is missing"]
    mk = re.search(r"<!-- traffic-table: (profiles/\S+) -->", text)
    if mk is None:
        return ["DESIGN.md has no traffic table (a line `<!-- traffic-table: profiles/<tag>_traffic.json -->` followed by `| kernel | GB |` rows)"]
    path = os.path.join(root, mk.group(1))
    if os.path.basename(path) != m["files"].get("traffic.json"):
        out.append(f"DESIGN.md's traffic table quotes {mk.group(1)}, the evidence set's traffic file is profiles/{m['files'].get('traffic.json')}")
    if not os.path.isfile(path):
        return out + [f"DESIGN.md's traffic table quotes {mk.group(1)}, which does not exist"]
    doc = json.load(open(path))
    kernels = {}
    for w in doc["workloads"].values():
        for k, v in w["kernels"].items():
            kernels.setdefault(k, v["hbm_bytes_per_launch"])
    rows = 0
    for ln in text[mk.end():].splitlines()[1:]:
        if not ln.startswith("|"):
            if rows: break
            continue
        cells = [c.strip().strip("`") for c in ln.strip("|").split("|")]
        if len(cells) < 2 or not cells[0].startswith("k_"):
            continue
        rows += 1
        try:
            gb = float(cells[1])
        except ValueError:
            out.append(f"DESIGN.md traffic table: `{ln.strip()}` has no number in its second column"); continue
        if cells[0] not in kernels:
            out.append(f"DESIGN.md traffic table: {cells[0]} is not in {mk.group(1)}"); continue
        have = kernels[cells[0]] / 1e9
        if rel(gb, have) > 0.05:
            out.append(f"DESIGN.md traffic table: {cells[0]} {gb} GB, {mk.group(1)} has {have:.2f} GB")
    if rows == 0:
        out.append("DESIGN.md's traffic table has no rows")
    return out


if __name__ == "__main__":
    p = check(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    print("ok" if not p else "\n".join(p))
    sys.exit(1 if p else 0)
