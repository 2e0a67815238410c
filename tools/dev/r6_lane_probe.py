"""Why are the lane-per-entry kernels slower on a mixed batch?  The same short entries as (a) a batch of their own inside the big text,
(b) the whole batch (long entries behind them, skipped by the lanes).  k_qv_hist id = k_qs_hist (+ k_qv_hist over the long ones in b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dextractor_amd import api, synth

n = 2_000_000
with api.Context(0) as ctx:
    movie = "m000_000"
    hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1
    lens = synth.lengths(n, 4242, sys.argv[1] if len(sys.argv) > 1 else "mixed_tail", 10000)
    ns = int((lens <= 4096).sum())
    hdr4 = synth.headers(n, 4242, lens, 0)
    rec = hlen + 5 * (lens.astype(np.uint64) + 1)
    off = (np.concatenate([[0], np.cumsum(rec)[:-1]]) + hlen).astype(np.uint64)
    tb = int(rec.sum())
    prof = synth.pacbio_profile()
    d_text = ctx.alloc(tb + 64)
    d_off, d_len = ctx.to_device(off), ctx.to_device(lens)
    d_hdr4, d_lut = ctx.to_device(hdr4.reshape(-1)), ctx.to_device(prof.table().reshape(-1))
    ctx.synth_quiva(4242, 0, n, d_off, d_len, d_hdr4, d_lut, prof.del_run, movie, d_text)
    ctx.sync()
    for what, m, tbytes in (("prefix of short entries, text_bytes of the whole", ns, tb + 64), ("prefix, text_bytes its own", ns, int(off[ns] - hlen)),
                            ("whole batch", n, tb + 64)):
        b = ctx.qv_batch(d_text, d_off, d_len, m, text_bytes=tbytes)
        p = ctx.qv_prescan(b)
        ctx.qv_hist(b, p)
        ctx.profile(True)
        for _ in range(3):
            ctx.qv_hist(b, p)
        ctx.sync()
        t = ctx.kernel_times()
        ctx.profile(False)
        print(what, m, {k: (round(v[0] / max(v[1], 1), 3), v[1]) for k, v in t.items() if v[1]})
