"""Debugging aid (GPU box): one encode with the sub-block index + one decode, with a watchdog and per-line diffs.

    python tools/dbg/dbg_sub.py [scratch|direct|text] [len,len,...]
"""
import sys, os, faulthandler
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
faulthandler.dump_traceback_later(40, exit=True)
import numpy as np
import _oracle as O
from dextractor_amd import _lib as L, api, synth
import test_gpu_parity as T
ctx = api.Context(0)
route = sys.argv[1] if len(sys.argv) > 1 else "scratch"
if route == "direct": os.environ["DEXGPU_DIRECT_ENCODE"] = "1"
if route == "text": os.environ["DEXGPU_NO_TOKENS"] = "1"
lens = np.array([int(x) for x in sys.argv[2].split(",")], np.uint32) if len(sys.argv) > 2 else np.array([7000] * 30, np.uint32)
c = synth.make_quiva(len(lens), seed=31, lens=lens)
n = len(c.len)
b, keep = T._upload_quiva(ctx, c)
p = ctx.qv_prescan(b)
hist, tot = ctx.qv_hist(b, p)
coding = api.qv_build(hist, tot, p, False)
ctx.qv_set_coding(coding, False)
blob, hoff, _ = api.frame_headers(c.hdr)
d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
cap = len(c.text) + 4096 * n + 4096
d_out = ctx.alloc(cap)
print("setup done", flush=True)
ctx.qv_subindex(True)
total = ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
print("encoded", total, flush=True)
img = np.zeros(len(c.text), np.uint8)
d_txt = ctx.to_device(img)
ctx.qv_decode(d_out, d_rec, d_hoff, d_seg, keep[2], n, True, d_txt, keep[1])
print("decoded", flush=True)
got = d_txt.download(np.uint8, len(c.text)).tobytes()
text = np.frombuffer(c.text, np.uint8)
bad = 0
for i in range(n):
    Ln, o = int(c.len[i]), int(c.off[i])
    for k in range(5):
        a0 = o + k * (Ln + 1)
        if got[a0:a0 + Ln + 1] != c.text[a0:a0 + Ln + 1]:
            g = np.frombuffer(got[a0:a0 + Ln + 1], np.uint8); w = text[a0:a0 + Ln + 1]
            d = np.nonzero(g != w)[0]
            print("entry", i, "L", Ln, "line", k, "first diff at", int(d[0]), "ndiff", len(d), flush=True)
            bad += 1
            if bad > 8: sys.exit(1)
print("OK" if not bad else "MISMATCH", flush=True)
