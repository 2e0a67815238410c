#!/usr/bin/env python3
"""After `gpurun -- bash tools/evidence.sh`: python tools/evidence_collect.py <tag>
Copies the evidence set from gpurun_out/ into profiles/ under the tag (bench lines, rocprofv3 kernel stats, the JSON lines of
the PROFILED runs themselves, PMC traffic, SQ counters, timeline, memory rates) and writes profiles/MANIFEST.json: the tag, the
commit and the hash of the device sources the set was measured on.  tests/test_profiles.py (profiles/check.py) holds the set
to that: a kernel edited since, or a rocprof average that disagrees with the bench line beside it, fails the CPU test suite."""
import hashlib, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import check                                              # noqa: E402  (device_sources_hash lives there: one definition)


def first_json_line(path):
    for ln in open(path):
        if ln.startswith("{"):
            return json.loads(ln)
    raise SystemExit(f"{path}: no JSON line")


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out")
    prof = os.path.join(ROOT, "profiles")
    files = {}
    for name, dst in (("ev_bench.json", "bench.json"), ("ev_bench_dexta.json", "bench_dexta.json"), ("ev_bench_dexar.json", "bench_dexar.json"),
                      ("ev_bench_pipeline.json", "bench_pipeline.json"), ("ev_bench_rank_of_8.json", "bench_rank_of_8.json"),
                      ("ev_cli_scale.json", "cli_scale.json"),
                      ("prof_stats.json", "profiled_bench.json"), ("prof_stats_dexta.json", "profiled_bench_dexta.json"),
                      ("prof_stats_dexar.json", "profiled_bench_dexar.json")):
        p = os.path.join(src, name)
        if os.path.isfile(p):
            json.dump(first_json_line(p), open(os.path.join(prof, f"{tag}_{dst}"), "w"))
            files[dst] = f"{tag}_{dst}"
    for name, dst in (("timeline.txt", "timeline.txt"), ("ev_copy_rate.txt", "copy_rate.txt"), ("ev_hbm_rates.txt", "hbm_rates.txt")):
        p = os.path.join(src, name)
        if os.path.isfile(p):
            shutil.copy(p, os.path.join(prof, f"{tag}_{dst}"))
            files[dst] = f"{tag}_{dst}"
    # the per-GPU figure an N > 1 bench line carries beside its own (bench.py: per_gpu_reference)
    p = os.path.join(prof, f"{tag}_bench_rank_of_8.json")
    if os.path.isfile(p):
        d = json.load(open(p))
        json.dump({"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "entries_per_gpu": d["config"]["entries_per_gpu"],
                   "mean_len": d["config"]["mean_len"], "roundtrip_bit_exact": d["roundtrip_bit_exact"], "n_gpus": 1,
                   "what": "one rank of the 8-GPU job (BASELINE configs[4]: 2.5 M entries x 10 kb, 125 GB, scratch budget 64 GB, index decode on) alone on its GPU",
                   "source": f"profiles/{tag}_bench_rank_of_8.json"}, open(os.path.join(prof, "per_gpu_reference.json"), "w"), indent=1)
        files["per_gpu_reference.json"] = "per_gpu_reference.json"
    # the CLI timing of the round as a text file too ("an r05 CLI timing file")
    p = os.path.join(prof, f"{tag}_cli_scale.json")
    if os.path.isfile(p):
        d = json.load(open(p))
        with open(os.path.join(prof, f"{tag}_cli_timing.txt"), "w") as f:
            f.write(f"# the tools end to end, tmpfs to tmpfs (tools/cli_scale.py): {d['file']}\n")
            for tool, runs in d["runs"].items():
                for r in (runs if isinstance(runs, list) else [runs]):
                    f.write(f"{tool}: " + ", ".join(f"{k} {v}" for k, v in r.items() if k != "marks_ms") + "\n")
                    for ms, what in r.get("marks_ms", []):
                        f.write(f"      {ms:9.1f} ms  {what}\n")
        files["cli_timing.txt"] = f"{tag}_cli_timing.txt"
    subprocess.check_call([sys.executable, os.path.join(prof, "summarize.py"), tag, src])
    for dst in ("kernel_stats.csv", "kernel_stats_dexta.csv", "kernel_stats_dexar.csv", "traffic.json", "sq_counters.json"):
        if os.path.isfile(os.path.join(prof, f"{tag}_{dst}")):
            files[dst] = f"{tag}_{dst}"
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"]).decode().strip()
    dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "dextractor_amd/csrc", "bench.py"]).decode().strip()
    manifest = {"tag": tag, "head": head, "tree_dirty_at_collection": bool(dirty),
                "device_sources_sha256": check.device_sources_hash(ROOT), "files": files}
    json.dump(manifest, open(os.path.join(prof, "MANIFEST.json"), "w"), indent=1)
    print(json.dumps(manifest, indent=1))
    problems = check.check(ROOT)
    print("check:", "ok" if not problems else problems)


if __name__ == "__main__":
    main()
