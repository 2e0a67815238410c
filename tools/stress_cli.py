"""Malformed text inputs through the packing tools and the reference's own binaries (oracle/_ref): tools/stress_cli.py [rounds] [seed].
A small valid .quiva / .fasta / .arrow gets one random edit (a line gone, doubled, cut short or grown, a header damaged, the file
truncated, an empty line, junk at the end); exit code, message and -- where both succeed -- the output file must be the same.  (GPU box.)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dextractor_amd import synth

BIN, REF = os.path.join(ROOT, "dextractor_amd", "bin"), os.path.join(ROOT, "oracle", "_ref")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
if not os.path.isfile(os.path.join(REF, "dexqv")):
    print("stress_cli: oracle/_ref not built: nothing to compare with"); sys.exit(0)

def edit(txt: bytes) -> (bytes, str):
    lines = txt.split(b"\n")[:-1]
    k = int(rng.integers(0, len(lines)))
    what = rng.choice(["drop", "double", "shorter", "longer", "header_mark", "header_rq", "header_slash", "header_num",
                       "truncate", "empty_line", "junk_tail", "no_final_newline", "last_plus_one", "none"])
    if what == "drop":       del lines[k]
    elif what == "double":   lines.insert(k, lines[k])
    elif what == "shorter":  lines[k] = lines[k][:-1]
    elif what == "longer":   lines[k] = lines[k] + lines[k][-1:] if lines[k] else b"a"
    elif what in ("header_mark", "header_rq", "header_slash", "header_num"):
        hs = [i for i, l in enumerate(lines) if l[:1] in (b"@", b">")]
        h = hs[int(rng.integers(0, len(hs)))]
        if what == "header_mark":    lines[h] = lines[h][1:]
        elif what == "header_rq":    lines[h] = lines[h].replace(b" RQ=", b" RX=").replace(b" SN=", b" SX=")
        elif what == "header_slash": lines[h] = lines[h].replace(b"/", b" ", 1)
        else:                        lines[h] = lines[h].replace(b"_", b"_x", 1)
    out = b"\n".join(lines) + b"\n"
    if what == "truncate":          out = txt[: int(rng.integers(1, len(txt)))]
    elif what == "empty_line":      out = b"\n".join(lines[:k] + [b""] + lines[k:]) + b"\n"
    elif what == "junk_tail":       out = txt + b"zzz"
    elif what == "no_final_newline": out = txt[:-1]
    elif what == "last_plus_one":   out = txt[:-1] + b"5"
    return out, what

bad = 0
for it in range(rounds):
    kind = rng.choice(["quiva", "fasta", "arrow"])
    n = int(rng.integers(2, 6))
    lens = rng.integers(3, 70, n).astype(np.uint32)
    seed = int(rng.integers(1, 1 << 30))
    txt = synth.make_quiva(n, seed=seed, lens=lens).text if kind == "quiva" else synth.make_seqfile(kind, n, seed=seed, lens=lens, width=int(rng.choice([20, 80]))).text
    data, what = edit(txt)
    tool, ext, oext = {"quiva": ("dexqv", ".quiva", ".dexqv"), "fasta": ("dexta", ".fasta", ".dexta"), "arrow": ("dexar", ".arrow", ".dexar")}[kind]
    res = []
    with tempfile.TemporaryDirectory() as d:
        for sub, exe in (("mine", os.path.join(BIN, tool)), ("ref", os.path.join(REF, tool))):
            dd = os.path.join(d, sub); os.mkdir(dd)
            open(os.path.join(dd, "x" + ext), "wb").write(data)
            r = subprocess.run([exe, "-k", "x"], cwd=dd, capture_output=True, timeout=120)
            o = os.path.join(dd, "x" + oext)
            res.append((r.returncode, r.stderr, open(o, "rb").read() if r.returncode == 0 and os.path.isfile(o) else None))
    if res[0] != res[1]:
        bad += 1
        os.makedirs(os.path.join(ROOT, "gpurun_out", "stress"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "stress", "cli_%d%s" % (bad, ext)), "wb").write(data)
        print(f"MISMATCH round {it} {tool} edit={what}: mine rc={res[0][0]} {res[0][1][:120]!r} out={None if res[0][2] is None else len(res[0][2])} | "
              f"ref rc={res[1][0]} {res[1][1][:120]!r} out={None if res[1][2] is None else len(res[1][2])}", flush=True)
    if it % 4 == 0:                                    # and the way back, with the unpacking tools' options, on the UNEDITED file
        flags = {"quiva": [["-k"], ["-k", "-U"]], "fasta": [["-k"], ["-k", "-U"], ["-k", "-w%d" % int(rng.choice([1, 17, 60, 100, 9999]))]],
                 "arrow": [["-k"], ["-k", "-w%d" % int(rng.choice([1, 17, 60, 100, 9999]))]]}[kind]
        fl = flags[int(rng.integers(0, len(flags)))]
        lossy = ["-l"] if kind == "quiva" and rng.random() < 0.3 else []
        res = []
        with tempfile.TemporaryDirectory() as d:
            for sub, bdir in (("mine", BIN), ("ref", REF)):
                dd = os.path.join(d, sub); os.mkdir(dd)
                open(os.path.join(dd, "x" + ext), "wb").write(txt)
                r1 = subprocess.run([os.path.join(bdir, tool), *lossy, "x"], cwd=dd, capture_output=True, timeout=120)       # (source removed: no -k)
                r2 = subprocess.run([os.path.join(bdir, "un" + tool), *fl, "x"], cwd=dd, capture_output=True, timeout=120)
                names = sorted(os.listdir(dd))
                res.append((r1.returncode, r1.stderr, r2.returncode, r2.stderr, names,
                            [open(os.path.join(dd, nm), "rb").read() for nm in names]))
        if res[0] != res[1]:
            bad += 1
            print(f"MISMATCH round {it} {tool} {lossy} / un{tool} {fl}: mine rc={res[0][0]},{res[0][2]} {res[0][1][:80]!r} {res[0][3][:80]!r} files={res[0][4]} | "
                  f"ref rc={res[1][0]},{res[1][2]} {res[1][1][:80]!r} {res[1][3][:80]!r} files={res[1][4]} same bytes: {res[0][5] == res[1][5]}", flush=True)
    if it % 20 == 19:
        print(f"{it + 1} rounds, {bad} mismatches", flush=True)
print("stress_cli:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
