"""Random shapes through the file drivers against the oracle (tests/_oracle.py: the C restatement, pinned to the reference):
tools/stress_files.py [rounds] [seed].  dexqv / undexqv, dexta / undexta, dexar / undexar -- bytes must be identical; where the
oracle refuses (or cannot read back) a file, the library must refuse it too.  (GPU box.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from dextractor_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = agree_refusals = 0

def lens_of(n):
    shape = rng.choice(["short", "mixed", "long", "tiny", "zeros"])
    if shape == "short":   l = rng.integers(0, 600, n)
    elif shape == "mixed": l = np.where(rng.random(n) < 0.3, rng.integers(0, 40, n), rng.integers(500, 30000, n))
    elif shape == "long":  l = rng.integers(20000, 90000, n)
    elif shape == "tiny":  l = rng.integers(0, 18, n)
    else:                  l = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 5000, n))
    return shape, l.astype(np.uint32)

def both(name, f_gpu, f_ora, dump=None):
    """(result or exception text) of both sides; a mismatch is counted and reported"""
    global bad, agree_refusals
    try: g = f_gpu()
    except Exception as e: g = e
    try: o = f_ora()
    except Exception as e: o = e
    if isinstance(o, Exception) and "one symbol" in str(o):       # (whatever the library does with such a file is acceptable here:
        agree_refusals += 1                                       #  refusing it, which is what it does, or decoding what can be decoded)
        return None
    if isinstance(g, Exception) and isinstance(o, Exception):
        agree_refusals += 1
        return None
    if isinstance(g, Exception) or isinstance(o, Exception) or g != o:
        bad += 1
        if dump is not None:
            os.makedirs(os.path.join(ROOT, "gpurun_out", "stress"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "stress", "files_%d.bin" % bad), "wb").write(dump)
        print(f"MISMATCH {name}: gpu {'raises ' + str(g)[:80] if isinstance(g, Exception) else len(g)}, "
              f"oracle {'raises ' + str(o)[:80] if isinstance(o, Exception) else len(o)}", flush=True)
        return None
    return g

with api.Context(0) as ctx:
    for it in range(rounds):
        n = int(rng.choice([1, 2, 3, 5, 8, 17, 40, 130]))
        shape, lens = lens_of(n)
        seed = int(rng.integers(1, 1 << 30))
        dp, sp = float(rng.choice([0.02, 0.3, 0.6, 0.85, 0.93, 0.97, 0.995])), float(rng.choice([0.02, 0.3, 0.6, 0.8, 0.95, 0.99]))
        lossy = bool(rng.random() < 0.25)
        c = synth.make_quiva(n, seed=seed, lens=lens, prof=synth.pacbio_profile(del_run_p=dp, sub_run_p=sp))
        tag = f"round {it} quiva n={n} {shape} del_p={dp} sub_p={sp} lossy={lossy} seed={seed}"
        dx = both(tag + " dexqv", lambda: ctx.dexqv(c.text, lossy), lambda: O.dexqv(c.text, lossy), c.text)
        if dx is not None:
            up = bool(rng.random() < 0.5)
            st = O.qv_scan(c.text)                           # a file with a stream of ONE symbol (the run character aside): its code has no
            hh = O.hist_array(st).copy()                     # bits; the reference -- and the oracle -- write it and return another text than
            if st.delChar >= 0: hh[0][st.delChar] = 0        # went in, or stop; the library refuses it
            if st.subChar >= 0: hh[3][st.subChar] = 0
            single = any(int(np.count_nonzero(hh[k])) == 1 for k in range(4))
            def ora_back():
                if single:
                    raise ValueError("a stream of one symbol: not a file the reference reads back")
                return O.undexqv(dx, upper=up)
            both(tag + " undexqv", lambda: ctx.undexqv(dx, upper=up), ora_back, c.text)
        for arrow in (False, True):
            wsrc = int(rng.choice([1, 13, 60, 80, 200]))
            t = synth.make_seqfile("arrow" if arrow else "fasta", n, seed=seed, lens=lens, width=wsrc, lower=bool(rng.random() < 0.3))
            tag2 = f"round {it} {'arrow' if arrow else 'fasta'} n={n} {shape} seed={seed}"
            px = both(tag2 + " pack", (lambda: ctx.dexar(t.text)) if arrow else (lambda: ctx.dexta(t.text)),
                      (lambda: O.dexar(t.text)) if arrow else (lambda: O.dexta(t.text)))
            if px is not None:
                w = int(rng.choice([1, 7, 15, 16, 17, 60, 80, 100, 1000]))
                both(tag2 + f" unpack w={w}", (lambda: ctx.undexar(px, w)) if arrow else (lambda: ctx.undexta(px, False, w)),
                     (lambda: O.undexar(px, w)) if arrow else (lambda: O.undexta(px, False, w)))
        if os.environ.get("STRESS_SHARDS") and it % 3 == 0:          # the same through 2 .. 4 contexts of this device (entry / read ranges)
            k = int(rng.integers(2, 5))
            ctxs = [ctx] + [api.Context(0) for _ in range(k - 1)]
            try:
                both(tag + f" dexqv over {k} contexts", lambda: api.dexqv_sharded(ctxs, c.text, lossy), lambda: O.dexqv(c.text, lossy), c.text)
                t = synth.make_seqfile("fasta", n, seed=seed, lens=lens, width=80)
                both(f"round {it} fasta over {k} contexts", lambda: api.pack2_sharded(ctxs, t.text), lambda: O.dexta(t.text))
            finally:
                for x in ctxs[1:]:
                    x.close()
        if it % 10 == 9:
            print(f"{it + 1} rounds, {bad} mismatches, {agree_refusals} refused by both", flush=True)
print("stress_files:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
