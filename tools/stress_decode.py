"""Random shapes through encode (with the group index) -> decode on the device, whole batches and random contiguous parts:
tools/stress_decode.py [rounds] [seed].  The decoded text must be the text that went in, byte for byte, and nothing may
be written outside the decoded part.  (GPU box; no oracle needed: lossless coding.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dextractor_amd import api, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = refused = 0
with api.Context(0) as ctx:
    for it in range(rounds):
        big = bool(os.environ.get("STRESS_BIG"))                    # thousands of entries: many units per ticket, every wave busy
        n = int(rng.choice([3000, 7000, 12000])) if big else int(rng.choice([1, 2, 3, 5, 8, 17, 40, 130, 400]))
        shape = rng.choice(["short", "tiny", "zeros"]) if big else rng.choice(["short", "mixed", "long", "tiny", "zeros"])
        if shape == "short":   lens = rng.integers(0, 600, n)
        elif shape == "mixed": lens = np.where(rng.random(n) < 0.3, rng.integers(0, 40, n), rng.integers(500, 30000, n))
        elif shape == "long":  lens = rng.integers(20000, 90000, n)
        elif shape == "tiny":  lens = rng.integers(0, 18, n)
        else:                  lens = np.where(rng.random(n) < 0.5, 0, rng.integers(1, 5000, n))
        lens = lens.astype(np.uint32)
        dp, sp = float(rng.choice([0.3, 0.6, 0.85, 0.93, 0.97, 0.995])), float(rng.choice([0.3, 0.6, 0.8, 0.95, 0.99]))
        c = synth.make_quiva(n, seed=int(rng.integers(1, 1 << 30)), lens=lens, prof=synth.pacbio_profile(del_run_p=dp, sub_run_p=sp))
        d_text = ctx.to_device(np.frombuffer(c.text, np.uint8))
        d_off, d_len = ctx.to_device(c.off), ctx.to_device(c.len)
        b = ctx.qv_batch(d_text, d_off, d_len, n, text_bytes=len(c.text))
        p = ctx.qv_prescan(b)
        hist, tot = ctx.qv_hist(b, p)
        hh = np.array(hist, dtype=np.uint64).reshape(6, 256).copy()   # a stream of ONE symbol (the run character aside): its code has no
        if p.delChar >= 0: hh[0][p.delChar] = 0                       # bits -- the reference writes such a file and cannot read it back
        if p.subChar >= 0: hh[3][p.subChar] = 0                       # (its undexqv stops, or returns a different text with exit code 0)
        single = any(int(np.count_nonzero(hh[s_])) == 1 for s_ in range(4))
        try:
            coding = api.qv_build(hist, tot, p, False)
        except Exception as e:                               # (a batch without symbols: the reference refuses it too)
            continue
        ctx.qv_set_coding(coding, False)
        blob, hoff, _ = api.frame_headers(c.hdr)
        d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
        d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
        cap = len(c.text) + 4096 * n + 4096
        d_out = ctx.alloc(cap)
        ctx.qv_subindex(True)
        try:
            ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
            parts = [(0, n)] + [tuple(sorted(rng.integers(0, n + 1, 2))) for _ in range(3)]
            for first, last in parts:
                count = int(last - first)
                if count <= 0:
                    continue
                first = int(first)
                img = np.frombuffer(c.text, np.uint8).copy()
                for i in range(n):
                    img[int(c.off[i]): int(c.off[i]) + 5 * (int(c.len[i]) + 1)] = 0
                d_txt = ctx.to_device(img)
                try:
                    ctx.qv_decode(d_out, d_rec.offset(8 * first), d_hoff.offset(8 * first), d_seg.offset(20 * first),
                                  d_len.offset(4 * first), count, True, d_txt, d_off.offset(8 * first))
                except Exception as e:                           # a stream of one symbol only (its code has no bits): refused, like
                    if single and ("in no table" in str(e) or "overruns its entry" in str(e)):   # the reference's own undexqv refuses the file
                        refused += 1
                        continue
                    os.makedirs("gpurun_out/stress", exist_ok=True)
                    open(f"gpurun_out/stress/round{it}.quiva", "wb").write(c.text)
                    print(f"UNEXPECTED refusal round {it}: n={n} shape={shape} del_p={dp} sub_p={sp} lens={list(map(int, c.len))[:12]}: {e}", flush=True)
                    bad += 1
                    continue
                got = d_txt.download(np.uint8, len(c.text)).tobytes()
                lo, hi = int(c.off[first]), int(c.off[first + count - 1]) + 5 * (int(c.len[first + count - 1]) + 1)
                ok = got[lo:hi] == c.text[lo:hi] and got[:lo] == img[:lo].tobytes() and got[hi:] == img[hi:].tobytes()
                if not ok and single:                           # (decoded although a stream has one symbol only -- that stream was
                    refused += 1                                # not needed for this part --, and differs: not a text the reference has)
                    continue
                if not ok:
                    bad += 1
                    a_, b_ = np.frombuffer(got, np.uint8), np.frombuffer(c.text, np.uint8).copy()
                    b_[:lo] = img[:lo]; b_[hi:] = img[hi:]
                    d = np.nonzero(a_ != b_)[0]
                    e = int(np.searchsorted(c.off, d[0], side="right") - 1)
                    rel = int(d[0]) - int(c.off[e])
                    Le = int(c.len[e])
                    ctx.qv_set_coding(coding, False)                # (the index is dropped: the lane-per-line kernels)
                    d_t2 = ctx.to_device(img)
                    ctx.qv_decode(d_out, d_rec.offset(8 * first), d_hoff.offset(8 * first), d_seg.offset(20 * first),
                                  d_len.offset(4 * first), count, True, d_t2, d_off.offset(8 * first))
                    g2 = d_t2.download(np.uint8, len(c.text)).tobytes()
                    print("   without the index:", "ok" if g2[lo:hi] == c.text[lo:hi] else "ALSO WRONG", flush=True)
                    print(f"MISMATCH round {it}: n={n} shape={shape} del_p={dp} sub_p={sp} part=({first},{count}) lens={list(map(int, c.len))[:12]} "
                          f"first diff at byte {int(d[0])} = entry {e} (L={Le}) line {rel // (Le + 1) if rel >= 0 else -1} col {rel % (Le + 1) if rel >= 0 else rel}, "
                          f"{len(d)} bytes differ; got {bytes(a_[d[:8]])!r} want {bytes(b_[d[:8]])!r}; delChar {p.delChar} subChar {p.subChar}", flush=True)
        finally:
            ctx.qv_subindex(False)
        if it % 10 == 9:
            print(f"{it + 1} rounds, {bad} mismatches, {refused} decodes refused (single-symbol streams)", flush=True)
print("stress_decode:", "OK" if bad == 0 else f"{bad} MISMATCHES")
sys.exit(1 if bad else 0)
