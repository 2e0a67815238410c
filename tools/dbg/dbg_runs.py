"""Debugging aid (GPU box): group-index decode of a bench-like batch with per-launch kernel times (rocprofv3-free)."""
import sys, os, faulthandler
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
faulthandler.dump_traceback_later(120, exit=True)
import numpy as np
from dextractor_amd import _lib as L, api, synth
import test_gpu_parity as T
ctx = api.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
c = synth.make_quiva(n, seed=5, dist="fixed", mean=10000)
b, keep = T._upload_quiva(ctx, c)
p = ctx.qv_prescan(b)
hist, tot = ctx.qv_hist(b, p)
coding = api.qv_build(hist, tot, p, False)
ctx.qv_set_coding(coding, False)
blob, hoff, _ = api.frame_headers(c.hdr)
d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
cap = len(c.text) // 2
d_out = ctx.alloc(cap)
ctx.qv_subindex(True)
total = ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
print("encoded", total, "params", p.delChar, p.subChar, flush=True)
d_txt = ctx.to_device(np.zeros(len(c.text), np.uint8))
for rep in range(2):
    ctx.profile(True)
    ctx.qv_decode(d_out, d_rec, d_hoff, d_seg, keep[2], n, True, d_txt, keep[1])
    print(ctx.kernel_times(), flush=True)
    ctx.profile(False)
got = d_txt.download(np.uint8, len(c.text)).tobytes()
data = lambda b_: [ln for ln in b_.split(b"\n") if not ln.startswith(b"@")]
print("OK" if data(got) == data(c.text) else "MISMATCH", flush=True)
