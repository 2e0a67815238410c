"""The six tools end to end on LARGE files, tmpfs to tmpfs (GPU box): tools/cli_scale.py [GB of .quiva, default 20] [--ref]
  dexqv / undexqv on an S GB .quiva (S / 50 KB entries of 10 kb, the bench's generator), dexta / undexta on the .fasta of the same
  reads; per run the wall time, the tool's own DEXGPU_TIMING marks (stages), and the effective rate; every round trip compared with
  its input (cmp); --ref: the reference's undexqv / undexta (oracle/_ref, one core) read the GPU tools' files back too.
One JSON object on stdout (bench.py picks it up as cpu_baseline.cli_end_to_end_large when it is in profiles/)."""
import atexit, json, os, re, shutil, signal, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dextractor_amd import api, synth

GB = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 20.0
REF = "--ref" in sys.argv
# a directory of this run's own (two runs on one box must not share files), removed however the run ends: the files are tens
# of GB of RAM-backed tmpfs.  --dir D: the caller made D and removes it itself as well (bench.py does, after a timeout's kill)
D = sys.argv[sys.argv.index("--dir") + 1] if "--dir" in sys.argv else tempfile.mkdtemp(prefix="cliscale.", dir="/dev/shm")
BIN = os.path.join(ROOT, "dextractor_amd", "bin")
REFBIN = os.path.join(ROOT, "oracle", "_ref")
os.makedirs(D, exist_ok=True)
atexit.register(shutil.rmtree, D, True)
for _sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(_sig, lambda *_: sys.exit(1))          # (sys.exit runs the atexit handler; SIGKILL cannot be caught: see --dir)


def free_shm():
    st = os.statvfs("/dev/shm")
    return st.f_bavail * st.f_frsize


mean, movie, seed = 10000, "m000_000", 20261003
hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1
n = max(1, int(GB * 1e9 / (hlen + 5 * (mean + 1))))


class Ptr:
    def __init__(self, t): self.t, self.ptr = t, t.data_ptr()


def log(*a): print(*a, file=sys.stderr, flush=True)


def make_files():
    """s.quiva by the bench's device generator; s.fasta: the same headers, random ACGT in 80-column lines"""
    lens = synth.lengths(n, seed, "fixed", mean)
    hdr4 = synth.headers(n, seed, lens, 0)
    rec = hlen + 5 * (lens.astype(np.uint64) + 1)
    off = (np.concatenate([[0], np.cumsum(rec)[:-1]]) + hlen).astype(np.uint64)
    total = int(rec.sum())
    prof = synth.pacbio_profile()
    with api.Context(0) as ctx:
        d_text = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.synth_quiva(seed, 0, n, Ptr(torch.from_numpy(off.view(np.int64)).cuda()), Ptr(torch.from_numpy(lens.view(np.int32)).cuda()),
                        Ptr(torch.from_numpy(hdr4.reshape(-1)).cuda()), Ptr(torch.from_numpy(prof.table().reshape(-1)).cuda()),
                        prof.del_run, movie, Ptr(d_text))
        ctx.sync(); torch.cuda.synchronize()
        with open(os.path.join(D, "s.quiva"), "wb") as f:
            for a in range(0, total, 1 << 30):
                f.write(d_text[a: min(total, a + (1 << 30))].cpu().numpy().tobytes())
        del d_text
    torch.cuda.empty_cache()
    letters = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    rows = mean // 80
    with open(os.path.join(D, "s.fasta"), "wb") as f:
        for a in range(0, n, 20000):
            b = min(n, a + 20000)
            body = letters[torch.randint(0, 4, (b - a, rows, 80), device="cuda")]
            body = torch.cat([body, torch.full((b - a, rows, 1), 10, dtype=torch.uint8, device="cuda")], 2).reshape(b - a, rows * 81).cpu().numpy()
            for i in range(a, b):
                f.write(synth.header_text("fasta", movie, hdr4[i], False))
                f.write(body[i - a].tobytes())
    return total


def run(tool, args, env=None):
    e = dict(os.environ, DEXGPU_TIMING="1", **(env or {}))
    t0 = time.perf_counter()
    r = subprocess.run([tool] + args, cwd=D, env=e, capture_output=True, timeout=1800)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"{tool} {args}: exit {r.returncode}: {r.stderr[-300:]}")
    marks = [(float(m.group(1)), m.group(2).strip()) for m in re.finditer(r"\[\w+\s+([\d.]+) ms\] (.*)", r.stderr.decode(errors="replace"))]
    return dt, marks


def same(a, b):
    return subprocess.run(["cmp", "-s", os.path.join(D, a), os.path.join(D, b)]).returncode == 0


res = {"file": {}, "runs": {}}
need = int(2.4 * GB * 1e9)                               # .quiva + its link's copy-free twin, the packed file, the unpacked text, the .fasta set
if free_shm() < need:
    print(json.dumps({"skipped": f"/dev/shm has {free_shm() >> 30} GiB free, the run wants {need >> 30}"}))
    sys.exit(0)
t0 = time.perf_counter()
qbytes = make_files()
fbytes = os.path.getsize(os.path.join(D, "s.fasta"))
res["file"] = {"quiva_bytes": qbytes, "fasta_bytes": fbytes, "entries": n, "mean_len": mean, "made_in_s": round(time.perf_counter() - t0, 1),
               "where": "tmpfs (/dev/shm) to tmpfs"}
log("files made:", res["file"])
for kind, pack, unpack, ext, px, uflags in (("quiva", "dexqv", "undexqv", ".quiva", ".dexqv", ["-U"]), ("fasta", "dexta", "undexta", ".fasta", ".dexta", ["-U", "-w80"])):
    size = qbytes if kind == "quiva" else fbytes
    os.replace(os.path.join(D, "s" + ext), os.path.join(D, "s0" + ext))
    os.link(os.path.join(D, "s0" + ext), os.path.join(D, "s" + ext))
    for rep in range(2):
        dt, marks = run(os.path.join(BIN, pack), ["-k", "s"])
        res["runs"].setdefault(pack, []).append({"s": round(dt, 3), "GBps": round(size / dt / 1e9, 2), "marks_ms": marks})
        log(pack, dt)
    if kind == "fasta":
        # ... and through a pipe (dexta -i: dx_file_pack2_stream, a chunk of whole records at a time): the same bytes, a chunk's memory
        t1 = time.perf_counter()
        with open(os.path.join(D, "s" + ext), "rb") as fi, open(os.path.join(D, "piped" + px), "wb") as fo:
            cat = subprocess.Popen(["cat"], stdin=fi, stdout=subprocess.PIPE)
            child = subprocess.Popen([os.path.join(BIN, pack), "-i"], stdin=cat.stdout, stdout=fo, stderr=subprocess.DEVNULL)
            # the tool's peak resident set: VmHWM of ITS address space, polled while it runs (wait4's ru_maxrss will not do: a child's
            # figure starts from what it held between fork and exec -- this Python process's gigabytes of torch, copy on write)
            hwm = 0
            while child.poll() is None:
                try:
                    with open("/proc/%d/status" % child.pid) as st:
                        for l in st:
                            if l.startswith("VmHWM"): hwm = max(hwm, int(l.split()[1]))
                except OSError:
                    pass
                time.sleep(0.02)
            cat.wait()
        dt = time.perf_counter() - t1
        res["runs"][pack + "_pipe"] = [{"s": round(dt, 3), "GBps": round(size / dt / 1e9, 2), "exit": child.returncode,
                                        "same_bytes_as_the_file_mode": child.returncode == 0 and same("piped" + px, "s" + px),
                                        "max_rss_MB": round(hwm / 1024, 1), "marks_ms": []}]
        log(pack, "-i", res["runs"][pack + "_pipe"])
        os.unlink(os.path.join(D, "piped" + px))
    os.unlink(os.path.join(D, "s" + ext))
    psize = os.path.getsize(os.path.join(D, "s" + px))
    for rep in range(2):
        dt, marks = run(os.path.join(BIN, unpack), ["-k"] + uflags + ["s"])
        # the text against the input: byte for byte (.fasta); the .quiva generator pads its header fields where undexqv prints %d
        # (undexqv.c:182), so there the text is packed once more and THAT file compared with the first (same tables, same records)
        if kind == "fasta":
            ok = same("s" + ext, "s0" + ext)
        elif rep == 1:
            os.replace(os.path.join(D, "s" + px), os.path.join(D, "first" + px))
            run(os.path.join(BIN, pack), ["-k", "s"])
            ok = same("s" + px, "first" + px)
            os.unlink(os.path.join(D, "first" + px))
        else:
            ok = None
        res["runs"].setdefault(unpack, []).append({"s": round(dt, 3), "text_GBps": round(size / dt / 1e9, 2), "round_trip_identical": ok, "marks_ms": marks})
        log(unpack, dt, ok)
        if REF and rep == 1 and os.path.isfile(os.path.join(REFBIN, unpack)):
            os.replace(os.path.join(D, "s" + ext), os.path.join(D, "ours" + ext))
            t1 = time.perf_counter()
            r = subprocess.run([os.path.join(REFBIN, unpack), "-k"] + uflags + ["s"], cwd=D, capture_output=True, timeout=3000)
            res["runs"]["reference_" + unpack] = {"s": round(time.perf_counter() - t1, 1), "exit": r.returncode, "cores": 1,
                                                  "its_text_is_the_gpu_tools_text": r.returncode == 0 and same("s" + ext, "ours" + ext)}
            log("reference", unpack, res["runs"]["reference_" + unpack])
            os.unlink(os.path.join(D, "ours" + ext))
        os.unlink(os.path.join(D, "s" + ext))
    res["file"][px[1:] + "_bytes"] = psize
    os.unlink(os.path.join(D, "s" + px)); os.unlink(os.path.join(D, "s0" + ext))
shutil.rmtree(D, True)
print(json.dumps(res))
