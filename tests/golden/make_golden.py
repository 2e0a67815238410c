#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference tools.

Run in the build container only (needs oracle/_ref, i.e. /root/reference compiled by
`make -C oracle ref`):

    python tests/golden/make_golden.py

For each case the script writes the input it built (seeded synthetic or hand-made edge cases)
and the bytes the reference produced for it:

    <case>.fasta  -> <case>.dexta  -> <case>.rt.fasta   (undexta, flags in CASES)
    <case>.arrow  -> <case>.dexar  -> <case>.rt.arrow   (undexar)
    <case>.quiva  -> <case>.dexqv  -> <case>.rt.quiva   (undexqv -U)
    qv_tiny.legacy.dexqv (older layout, derived from qv_tiny.dexqv) -> qv_tiny.legacy.rt.quiva (undexqv -U),
                                                                         qv_tiny.legacy.rt_lower.quiva (undexqv)

Large inputs are stored gzip-compressed; for the 10 MB BASELINE config-1 corpus only the seed and
the SHA-256 of the reference's outputs are stored (hashes.json).  Fixtures are data: inputs and
expected outputs only -- no reference source is copied.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dextractor_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GZ_ABOVE = 64 * 1024


def run_tool(tool, flags, src_bytes, src_ext, dst_ext):
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "x" + src_ext)
        with open(src, "wb") as f:
            f.write(src_bytes)
        r = subprocess.run([os.path.join(REF, tool), "-k", *flags, src], capture_output=True)
        if r.returncode != 0:
            raise RuntimeError(f"{tool} failed: {r.stderr.decode()}")
        with open(os.path.join(d, "x" + dst_ext), "rb") as f:
            return f.read()


def store(name, data):
    path = os.path.join(HERE, name)
    for p in (path, path + ".gz"):
        if os.path.exists(p):
            os.remove(p)
    if len(data) > GZ_ABOVE:
        with open(path + ".gz", "wb") as f:
            f.write(gzip.compress(data, 9, mtime=0))
    else:
        with open(path, "wb") as f:
            f.write(data)


def store_rt(name, rt, txt):
    """Round-trip text: stored only when it differs from the input (returns True if identical)."""
    for p in (os.path.join(HERE, name), os.path.join(HERE, name + ".gz")):
        if os.path.exists(p):
            os.remove(p)
    if rt == txt:
        return True
    store(name, rt)
    return False


# ------------------------------------------------------------------------------------------
#  hand-made edge cases (SURVEY.md 8(c) fixture list, Appendix B)
# ------------------------------------------------------------------------------------------

def fasta_edge():
    recs = [
        (">mv/5/0_5 RQ=0.851", "ACGTN"),                  # Appendix B vector 1
        (">mv/5/10_10 RQ=0.800", ""),                     # L = 0, delta 0
        (">mv/260/3_10", "acgtacg"),                      # delta 255, missing RQ, lower case, L%4=3
        (">mv/770/0_2 RQ=0.085", "GT"),                   # delta 510, leading-zero RQ
        (">mv/771/0_4 RQ=0.9", "TTTT"),                   # delta 1, L%4=0
        (">mv/1025/0_1 RQ=0.75", "C"),                    # delta 254, L%4=1
        (">mv/1281/0_6 RQ=0.123456", "nNaAxX"),           # delta 256, non-ACGT letters -> 0
        (">mv/1881/7_180 RQ=0.77", "ACGT" * 43 + "G"),    # delta 600, multi-line (173 = 2 full + 13)
        (">mv/1881/0_80 RQ=0.5", "T" * 80),               # exactly one full line
    ]
    out = []
    for h, s in recs:
        out.append(h)
        out += [s[i:i + 80] for i in range(0, len(s), 80)]
    return ("\n".join(out) + "\n").encode()


def arrow_edge():
    recs = [
        (">mv/5/0_6 SN=6.81,99.99,100.50,0.00", "123405"),      # Appendix B vector 2
        (">mv/5/0_0 SN=1.00,2.00,3.00,4.00", ""),
        (">mv/300/2_9 SN=12.34,56.78,9.10,11.12", "4321G1A"),   # 'G' quirk -> 2, 'A' -> 3
        (">mv/555/0_165 SN=99.98,0.01,50.50,7.07", "1234" * 41 + "2"),
    ]
    out = []
    for h, s in recs:
        out.append(h)
        out += [s[i:i + 80] for i in range(0, len(s), 80)]
    return ("\n".join(out) + "\n").encode()


def quiva_from_lines(entries, movie="m000_000"):
    """entries: list of (well, beg, qv, [5 byte-strings of equal length])"""
    out = []
    for well, beg, qv, lines in entries:
        L = len(lines[0])
        assert all(len(x) == L for x in lines)
        out.append(f"@{movie}/{well}/{beg}_{beg + L} RQ=0.{qv}\n".encode())
        out += [bytes(x) + b"\n" for x in lines]
    return b"".join(out)


def fib_stream(nsym, base=40):
    """Symbols with Fibonacci counts: forces a Huffman code longer than 16 bits (type-2 scheme)."""
    f = [1, 1]
    while len(f) < nsym:
        f.append(f[-1] + f[-2])
    s = np.concatenate([np.full(c, base + i, np.uint8) for i, c in enumerate(f)])
    rng = np.random.Generator(np.random.PCG64(7))
    rng.shuffle(s)
    return s


def quiva_type2():
    """ins stream with 22 Fibonacci-weighted symbols => ins scheme is truncated (type 2, 8-bit
    escapes); totChar = 46367 < 100000 so no substitution run char."""
    ins = fib_stream(22)
    L = len(ins)
    body = synth.qv_lines(99, 0, L, synth.pacbio_profile())
    body[2] = ins
    ents, at, well = [], 0, 3
    for ln in (9000, 12000, 7000, L - 28000):
        ents.append((well, 10, 801, [body[r, at:at + ln].tobytes() for r in range(5)]))
        at += ln
        well += 300
    return quiva_from_lines(ents)


def quiva_runs():
    """Deletion runs of 254/255/256/300/700/1000/3000; entries ending in / not ending in the run
    char; delChar first seen in the third entry; an entry that is one single run; an entry with a
    run of exactly 65535."""
    prof = synth.pacbio_profile()
    rng = np.random.Generator(np.random.PCG64(11))

    def entry(del_vals, seedno):
        L = len(del_vals)
        b = synth.qv_lines(1234, seedno, L, prof)
        b[0] = del_vals
        b[1] = prof.tag_lut[rng.integers(0, 4096, L)]
        b[1][b[0] == ord("2")] = ord("N")
        return [b[r].tobytes() for r in range(5)]

    def runs_then(lens_, tail):
        parts = []
        for r in lens_:
            parts.append(np.full(r, ord("2"), np.uint8))
            parts.append(np.array([int(rng.integers(34, 50))], np.uint8))
        parts.append(np.full(tail, ord("2"), np.uint8))
        return np.concatenate(parts)

    nodel = rng.integers(34, 50, 400).astype(np.uint8)          # no run char, no 'N' tag
    ents = [
        (4, 0, 812, entry(nodel, 0)),
        (9, 5, 700, entry(rng.integers(34, 50, 37).astype(np.uint8), 1)),
        (260, 0, 850, entry(runs_then([0, 1, 254, 255, 256, 300], 0), 2)),    # delChar found here
        (261, 0, 851, entry(runs_then([700, 1000, 3000, 2], 5), 3)),          # ends in the run char
        (600, 0, 852, entry(np.full(777, ord("2"), np.uint8), 4)),            # one single run
        (601, 0, 853, entry(runs_then([65535, 3], 1), 5)),
        (900, 0, 854, entry(runs_then([5, 5, 5], 0), 6)),                     # ends in a non-run symbol
    ]
    return quiva_from_lines(ents)


CASES = []


def legacy_pack2(O):
    """Older / other-endian .dexta and .dexar layouts (undexta.c:140-159, 211-240; undexar.c:138-145).  No current tool
    writes them: the images are derived from the reference's own ta_small.dexta / ar_small.dexar (fields narrowed,
    bytes swapped: O.rewrite_pack2) and the expected text is what the REAL reference undexta / undexar print."""
    ta = O.golden("ta_small.dexta")
    for tag, kw in (("legacy", dict(legacy=True)), ("swapped", dict(swap=True)), ("legacy_swapped", dict(legacy=True, swap=True))):
        img = O.rewrite_pack2(ta, **kw)
        store(f"ta_small.{tag}.dexta", img)
        store(f"ta_small.{tag}.rt.fasta", run_tool("undexta", ["-U"], img, ".dexta", ".fasta"))
    ar = O.rewrite_pack2(O.golden("ar_small.dexar"), arrow=True, swap=True)
    store("ar_small.swapped.dexar", ar)
    store("ar_small.swapped.rt.arrow", run_tool("undexar", [], ar, ".dexar", ".arrow"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "legacy_pack2":          # (only these fixtures; the others stay as they are)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        return legacy_pack2(O)
    if not os.path.isdir(REF):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` in the build container first")
    hashes = {}

    # ---- dexta / undexta ----
    fa = {
        "ta_edge": (fasta_edge(), []),
        "ta_small": (synth.make_seqfile("fasta", 12, seed=101, mean=900).text, ["-U"]),
        "ta_lower_w60": (synth.make_seqfile("fasta", 5, seed=102, mean=700, width=60, lower=True).text, ["-w60"]),
    }
    for name, (txt, uflags) in fa.items():
        dx = run_tool("dexta", [], txt, ".fasta", ".dexta")
        rt = run_tool("undexta", uflags, dx, ".dexta", ".fasta")
        store(name + ".fasta", txt); store(name + ".dexta", dx)
        CASES.append({"name": name, "kind": "fasta", "undex_flags": uflags, "rt_is_input": store_rt(name + ".rt.fasta", rt, txt)})

    # ---- dexar / undexar ----
    ar = {
        "ar_edge": (arrow_edge(), []),
        "ar_small": (synth.make_seqfile("arrow", 12, seed=103, mean=900).text, []),
    }
    for name, (txt, uflags) in ar.items():
        dx = run_tool("dexar", [], txt, ".arrow", ".dexar")
        rt = run_tool("undexar", uflags, dx, ".dexar", ".arrow")
        store(name + ".arrow", txt); store(name + ".dexar", dx)
        CASES.append({"name": name, "kind": "arrow", "undex_flags": uflags, "rt_is_input": store_rt(name + ".rt.arrow", rt, txt)})

    # ---- dexqv / undexqv ----
    nodel = synth.pacbio_profile()
    nodel.del_run = -1                      # tag never 'N' -> delChar stays -1
    qv = {
        # regime (i): totChar < 100000: no substitution run char is ever chosen
        "qv_tiny": (synth.make_quiva(6, seed=201, mean=600).text, []),
        # regime (ii): 100000 <= totChar < 200000: chosen at the scan, dropped by Create_QVcoding
        "qv_mid": (synth.make_quiva(14, seed=202, mean=9000).text, []),
        # regime (iii): totChar >= 200000: both run schemes active
        "qv_full": (synth.make_quiva(24, seed=203, mean=10000).text, []),
        # regime (iv): lossy (same input as qv_full)
        "qv_lossy": (synth.make_quiva(24, seed=203, mean=10000).text, ["-l"]),
        # no 'N' tag anywhere: delChar == -1, plain deletion coding, full tag line packed
        "qv_nodel": (synth.make_quiva(5, seed=204, mean=800, prof=nodel).text, []),
        "qv_type2": (quiva_type2(), []),
        "qv_runs": (quiva_runs(), []),
    }
    for name, (txt, flags) in qv.items():
        dx = run_tool("dexqv", flags, txt, ".quiva", ".dexqv")
        rt = run_tool("undexqv", ["-U"], dx, ".dexqv", ".quiva")
        if name != "qv_lossy":
            store(name + ".quiva", txt)
        store(name + ".dexqv", dx)
        CASES.append({"name": name, "kind": "quiva", "flags": flags,
                      "input": "qv_full" if name == "qv_lossy" else name,
                      "rt_is_input": store_rt(name + ".rt.quiva", rt, txt)})

    # ---- older .dexqv layout (no 0x55aa key, uint16 beg/end/qv: undexqv.c:104-109, 159-179) ----
    # No current tool writes it; the image is derived from the reference's own qv_tiny.dexqv (fields
    # narrowed, key dropped) and the expected text is what the REAL reference undexqv prints for it.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O                                  # noqa: E402  (only its byte-shuffling helper)
    from dextractor_amd import api                       # noqa: E402  (host-side walk: record boundaries)
    dx = run_tool("dexqv", [], qv["qv_tiny"][0], ".quiva", ".dexqv")
    leg = O.legacy_dexqv(dx, api.qv_walk(dx))
    store("qv_tiny.legacy.dexqv", leg)
    store("qv_tiny.legacy.rt.quiva", run_tool("undexqv", ["-U"], leg, ".dexqv", ".quiva"))
    store("qv_tiny.legacy.rt_lower.quiva", run_tool("undexqv", [], leg, ".dexqv", ".quiva"))

    legacy_pack2(O)

    # ---- BASELINE config 1: 1000 reads, mean 10 kb -- hashes only ----
    c1 = synth.make_seqfile("fasta", 1000, seed=20261003, mean=10000)
    dx = run_tool("dexta", [], c1.text, ".fasta", ".dexta")
    rt = run_tool("undexta", ["-U"], dx, ".dexta", ".fasta")
    assert rt == c1.text, "reference round trip of config 1 is not byte-identical"
    hashes["config1_fasta"] = {"seed": 20261003, "n": 1000, "mean": 10000,
                               "input_sha256": hashlib.sha256(c1.text).hexdigest(),
                               "dexta_sha256": hashlib.sha256(dx).hexdigest(),
                               "dexta_bytes": len(dx), "input_bytes": len(c1.text)}
    q1 = synth.make_quiva(1000, seed=20261003, mean=10000)
    for lossy in (0, 1):
        dx = run_tool("dexqv", ["-l"] if lossy else [], q1.text, ".quiva", ".dexqv")
        hashes["config4s_quiva" + ("_lossy" if lossy else "")] = {
            "seed": 20261003, "n": 1000, "mean": 10000,
            "input_sha256": hashlib.sha256(q1.text).hexdigest(),
            "dexqv_sha256": hashlib.sha256(dx).hexdigest(),
            "dexqv_bytes": len(dx), "input_bytes": len(q1.text)}

    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(CASES, f, indent=1)
    with open(os.path.join(HERE, "hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(HERE, x)) for x in os.listdir(HERE))
    print(f"wrote {len(CASES)} cases, {tot / 1024:.0f} KiB in {HERE}")


if __name__ == "__main__":
    main()
