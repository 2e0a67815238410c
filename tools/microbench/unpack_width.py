"""k_pack2_decode against the line width: tools/microbench/unpack_width.py [reads] -- any bytes are valid 2-bit codes, so the
packed input is noise; widths below 16 take the per-byte path (several line ends may fall into a lane's 16 bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dextractor_amd import api, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
Ln = 10_000
with api.Context(0) as ctx:
    clen = (Ln + 3) // 4
    rng = np.random.default_rng(5)
    tile = rng.integers(0, 256, size=clen * 4096, dtype=np.uint8)
    packed = np.tile(tile, (n + 4095) // 4096)[: n * clen]
    d_in = ctx.to_device(packed)
    d_ioff = ctx.to_device((np.arange(n, dtype=np.uint64) * clen))
    d_nsym = ctx.to_device(np.full(n, Ln, np.uint32))
    for w in (8, 15, 16, 17, 40, 60, 80, 100, 1000, 10000):
        T = Ln + (Ln + w - 1) // w
        d_ooff = ctx.to_device(np.arange(n, dtype=np.uint64) * T)
        d_out = ctx.alloc(n * T + 64)
        ctx.pack2_decode(L.DX_LETTERS_UPPER, d_in, d_ioff, d_nsym, n, w, d_out, d_ooff)
        ctx.sync()
        ctx.profile(True)
        for _ in range(3):
            ctx.pack2_decode(L.DX_LETTERS_UPPER, d_in, d_ioff, d_nsym, n, w, d_out, d_ooff)
        ctx.sync()
        t = ctx.kernel_times()["k_pack2_decode"]
        ctx.profile(False)
        ms = t[0] / t[1]
        print(f"width {w:5d}: {ms:8.3f} ms per launch, {(n * T + n * clen) / ms / 1e6:7.1f} GB/s of traffic", flush=True)
        del d_out, d_ooff
