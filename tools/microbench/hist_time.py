"""Time k_qv_hist (and optionally the encoder) alone on a synthetic batch: tools/microbench/hist_time.py [entries] [reps]
Used for perturbation experiments (library variants built with parts of a kernel compiled out; DEXGPU_LIB selects one):
the results of such variants are wrong on purpose, only the kernel time is of interest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from dextractor_amd import api, synth

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 1_000_000
reps = int(args[1]) if len(args) > 1 else 5
with api.Context(0) as ctx:
    movie = "m000_000"
    hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1
    mean = int(args[2]) if len(args) > 2 else 10000
    lens = synth.lengths(n, 4242, "fixed", mean)
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
    b = ctx.qv_batch(d_text, d_off, d_len, n, text_bytes=tb + 64)
    p = ctx.qv_prescan(b)
    ctx.qv_hist(b, p)
    enc = "--encode" in sys.argv
    if enc:
        hist, tot = ctx.qv_hist(b, p)
        coding = api.qv_build(hist, tot, p, False)
        ctx.qv_set_coding(coding, False)
        blob, hoff, _ = api.frame_headers(hdr4, None, 0)
        d_hdr, d_hoff = ctx.to_device(blob.copy()), ctx.to_device(hoff)
        d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
        cap = int(hoff[-1]) + api.qv_out_bound(hist, n, coding, False) + 4096
        d_out = ctx.alloc(cap)
        ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
    ctx.profile(True)
    for _ in range(reps):
        if enc:
            ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
        else:
            ctx.qv_hist(b, p)
    ctx.sync()
    t = ctx.kernel_times()
    print(os.path.basename(os.environ.get("DEXGPU_LIB", "main")), {k: round(v[0] / max(v[1], 1), 3) for k, v in t.items() if v[1]})
