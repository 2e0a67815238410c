"""Device walk against the host walk on a synthetic file: tools/dev/walk_try.py [entries] [mean] [piece]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dextractor_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
mean = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
if len(sys.argv) > 3:
    os.environ["DEXGPU_TEST"] = "walk_piece=" + sys.argv[3]
with api.Context(0) as ctx:
    c = synth.make_quiva(n, seed=77, mean=mean)
    img = ctx.dexqv(c.text)
    print("image", len(img), "bytes,", n, "entries", flush=True)
    t0 = time.time(); h = api.qv_walk(img); t1 = time.time()
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    first = 2 + used
    d = ctx.to_device(np.frombuffer(img, np.uint8))
    ctx.profile(True)
    t2 = time.time(); x = ctx.qv_walk_device(d, len(img), first, coding, 1, flip); t3 = time.time()
    g = x.download()
    print("host walk %.3f s, device walk %.3f s, pieces %d of %d bytes" % (t1 - t0, t3 - t2, x.pieces, x.piece_bytes), ctx.kernel_times().get("k_qv_walk"))
    ok = g["n"] == h["n"]
    for k in ("rec_off", "hdr_off", "seg", "len", "hdr4"):
        same = g[k].shape == h[k].shape and bool((g[k] == h[k]).all())
        ok = ok and same
        if not same:
            bad = np.argwhere(g[k] != h[k])[:3] if g[k].shape == h[k].shape else None
            print("DIFF", k, g[k].shape, h[k].shape, bad)
    print("identical" if ok else "DIFFERENT", g["n"], h["n"])
    x.free()
