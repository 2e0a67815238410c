"""Random batches of short entries with long ones among them through the file driver (dx_file_dexqv: the lanes for the short entries, the
wave-per-entry kernels for the long ones as a batch of their own) against the oracle, byte for byte, and back through undexqv.
usage (GPU box): python tools/stress_mixed.py [rounds] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
from dextractor_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
routes = {}
with api.Context(0) as ctx:
    for it in range(rounds):
        rng = np.random.Generator(np.random.PCG64(seed0 * 1000 + it))
        n = int(rng.integers(4100, 9000))
        frac = float(rng.choice([0.0, 0.002, 0.02, 0.1, 0.3]))
        smax = int(rng.choice([40, 300, 700, 1500, 3000]))
        short = rng.integers(0 if rng.random() < 0.3 else 1, smax + 1, n)
        long_ = np.clip(rng.lognormal(np.log(float(rng.choice([3000, 9000, 30000]))), 0.5, n), 500, 90000).astype(np.int64)
        lens = np.where(rng.random(n) < frac, long_, short).astype(np.uint32)
        dens = float(rng.choice([0.3, 0.6, 0.8, 0.85, 0.95, 0.995]))
        prof = synth.pacbio_profile(dens, float(rng.choice([0.3, 0.8, 0.97])))
        lossy = bool(rng.random() < 0.25)
        flags = []
        if rng.random() < 0.5: flags.append("short_force")
        if rng.random() < 0.15: flags.append("sizes_from_tokens")
        if rng.random() < 0.1: flags.append("no_tokens")
        if rng.random() < 0.2: flags.append("short_cut=%d" % int(rng.choice([64, 256, 1024, 4096])))
        os.environ["DEXGPU_TEST"] = ",".join(flags)
        c = synth.make_quiva(n, seed=seed0 * 77 + it, lens=lens, prof=prof)
        want = O.dexqv(c.text, lossy)
        got = ctx.dexqv(c.text, lossy)
        r = ctx.qv_onepass_info()["direct"]
        routes[r] = routes.get(r, 0) + 1
        assert got == want, ("dexqv", it, n, frac, smax, dens, lossy, flags, r)
        if not lossy:
            back = ctx.undexqv(got, upper=True)
            assert back.split(b"\n")[1::6] == c.text.split(b"\n")[1::6], ("undexqv", it)
        print(it, n, frac, smax, dens, lossy, flags, "route", r, len(c.text), "->", len(got), flush=True)
print("clean:", rounds, "rounds; routes taken", routes)
