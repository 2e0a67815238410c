"""The routes dx_qv_encode_onepass takes under memory pressure, at a size where they matter (needs an MI355X).

BASELINE configs[4] gives every GPU a 2.5 M-entry slice (125 GB of QV bytes): beside it, its output and its tokens
there is room for only a fraction of the scratch a 1 M-entry batch gets, so the encoder works in more, smaller
groups or, with no room at all, sizes first and in place.  Here the same pressure is put on a batch that fits a test
(dx_set_scratch_budget scales the memory the encoder may assume down by the same factor): every route must write the
same bytes, and those bytes are pinned to the oracle on a sample."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from dextractor_amd import _lib as L
from dextractor_amd import api, synth

pytestmark = pytest.mark.gpu

N, MEAN, SEED = 120_000, 10_000, 4242          # 6 GB of QV bytes; the scratch slots of the whole batch: ~5 GB


class Corpus:
    def __init__(self, ctx, n=N, mean=MEAN, seed=SEED, dist="lognormal"):
        movie = "m000_000"
        self.hlen = hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1
        self.lens = lens = synth.lengths(n, seed, dist, mean)
        self.hdr4 = hdr4 = synth.headers(n, seed, lens, 0)
        rec = hlen + 5 * (lens.astype(np.uint64) + 1)
        self.off = off = (np.concatenate([[0], np.cumsum(rec)[:-1]]) + hlen).astype(np.uint64)
        self.text_bytes = int(rec.sum())
        prof = synth.pacbio_profile()
        self.d_text = ctx.alloc(self.text_bytes + 64)
        self.d_off, self.d_len = ctx.to_device(off), ctx.to_device(lens)
        d_hdr4, d_lut = ctx.to_device(hdr4.reshape(-1)), ctx.to_device(prof.table().reshape(-1))
        ctx.synth_quiva(seed, 0, n, self.d_off, self.d_len, d_hdr4, d_lut, prof.del_run, movie, self.d_text)
        ctx.sync()
        self.batch = ctx.qv_batch(self.d_text, self.d_off, self.d_len, n, text_bytes=self.text_bytes + 64)
        blob, hoff, _ = api.frame_headers(hdr4, None, 0)
        self.hoff = hoff
        self.d_hdr, self.d_hoff = ctx.to_device(blob.copy()), ctx.to_device(hoff)
        self.d_rec, self.d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
        self.n = n

    def encode(self, ctx, budget):
        """scan + tables + one-pass encode under `budget` bytes of scratch; -> (stream bytes, route)"""
        ctx.set_scratch_budget(budget)
        p = ctx.qv_prescan(self.batch)
        hist, tot = ctx.qv_hist(self.batch, p)
        coding = api.qv_build(hist, tot, p, False)
        ctx.qv_set_coding(coding, False)
        cap = int(self.hoff[-1]) + api.qv_out_bound(hist, self.n, coding, False) + 4096
        d_out = ctx.alloc(cap)
        total = ctx.qv_encode_onepass(self.batch, self.d_hdr, self.d_hoff, self.d_seg, self.d_rec, d_out, cap)
        info = ctx.qv_onepass_info()
        out = d_out.download(np.uint8, total)
        d_out.free()
        return out, info, coding


def test_onepass_routes_under_memory_pressure_write_the_same_stream():
    with api.Context(0) as ctx:
        c = Corpus(ctx)
        ref, info0, coding = c.encode(ctx, 0)                       # no budget: by free device memory
        assert info0["direct"] == 0 and info0["groups"] >= 1 and info0["tokens"] == 1
        # the first 300 records against the oracle (the same entries as a small file give the same tables only if the
        # whole corpus does: so compare record by record with the oracle's entry encoder under THESE tables)
        rec = c.d_rec.download(np.uint64, c.n + 1)
        text = c.d_text.download(np.uint8, int(c.off[300]))
        oc = O.Coding()                                             # (same layout as dx_qv_coding)
        C.memmove(C.byref(oc), C.byref(coding), C.sizeof(coding))
        for i in range(299):
            o, ln = int(c.off[i]), int(c.lens[i])
            lines = np.stack([text[o + k * (ln + 1): o + k * (ln + 1) + ln] for k in range(5)])
            want, _ = O.qv_encode_entry(oc, False, lines)
            hl = int(c.hoff[i + 1] - c.hoff[i])
            assert ref[int(rec[i]) + hl: int(rec[i + 1])].tobytes() == want, i
        seen = {(info0["groups"], info0["direct"])}
        # a half, an eighth, an eighteenth of what the slots of the whole batch take (three regions of a group's slots
        # must fit: 6, 24, 54 groups), then nothing to speak of (no slots at all: sizes first, records in place)
        whole = info0["groups"] * info0["region_bytes"]
        for budget in (whole // 2 + 28 * c.n, whole // 8 + 28 * c.n, whole // 18 + 28 * c.n, 80 << 20):
            got, info, _ = c.encode(ctx, budget)
            assert info["avail_bytes"] == budget
            assert len(got) == len(ref) and (got == ref).all(), info
            seen.add((info["groups"], info["direct"]))
        assert any(d == 1 for _, d in seen), seen                   # the sizes-first route ran ...
        assert len({g for g, d in seen if d == 0}) >= 3, seen       # ... and three different groupings of the slot route
        ctx.set_scratch_budget(0)


def test_budget_env_overrides_and_route_is_reported(monkeypatch):
    with api.Context(0) as ctx:
        c = Corpus(ctx, n=3000, mean=4000)
        a, info_a, _ = c.encode(ctx, 0)
        monkeypatch.setenv("DEXGPU_SCRATCH_BUDGET", str(16 << 20))
        b, info_b, _ = c.encode(ctx, 0)
        assert info_b["avail_bytes"] == 16 << 20
        assert (a == b).all()
        small = synth.make_quiva(40, seed=5, mean=3000)
        assert ctx.dexqv(small.text) == O.dexqv(small.text)         # the file driver under the same budget


def test_chained_placement_writes_the_same_stream(monkeypatch):
    """DEXGPU_CHAIN: sizes first, record offsets by a decoupled look-back over per-entry status words, records written in
    place -- no scratch slots, no compaction.  Same bytes as the slot route, on ragged lengths, with entries whose
    tokens cannot be used in the chain (bytes >= 128 in a run-coded line: their sizes and records come from the
    text-reading kernels) and with a d_out that is too small (DX_E_SPACE, nothing overrun)."""
    with api.Context(0) as ctx:
        c = Corpus(ctx, n=60_000, mean=6000)
        ref, info0, _ = c.encode(ctx, 0)
        monkeypatch.setenv("DEXGPU_CHAIN", "1")
        got, info, _ = c.encode(ctx, 0)
        assert info["direct"] == 2 and info["groups"] == 0
        assert len(got) == len(ref) and (got == ref).all()
        rec_chain = c.d_rec.download(np.uint64, c.n + 1)
        seg_chain = c.d_seg.download(np.uint32, 5 * c.n)
        monkeypatch.delenv("DEXGPU_CHAIN")
        c.encode(ctx, 0)
        assert (c.d_rec.download(np.uint64, c.n + 1) == rec_chain).all()       # the same index beside the stream
        assert (c.d_seg.download(np.uint32, 5 * c.n) == seg_chain).all()
        # a too small d_out: reported, nothing beyond it touched
        monkeypatch.setenv("DEXGPU_CHAIN", "1")
        p = ctx.qv_prescan(c.batch)
        hist, tot = ctx.qv_hist(c.batch, p)
        ctx.qv_set_coding(api.qv_build(hist, tot, p, False), False)
        cap = len(ref) // 2
        d_out = ctx.to_device(np.full(cap + 4096, 0xEE, np.uint8))
        with pytest.raises(L.DexGPUError) as e:
            ctx.qv_encode_onepass(c.batch, c.d_hdr, c.d_hoff, c.d_seg, c.d_rec, d_out, cap)
        assert e.value.code == -8
        assert (d_out.download(np.uint8, 4096, offset=cap) == 0xEE).all()


def test_chained_placement_small_files_and_text_entries(monkeypatch):
    monkeypatch.setenv("DEXGPU_CHAIN", "1")
    with api.Context(0) as ctx:
        for seed, n, mean in ((3, 1, 50), (4, 7, 3000), (5, 300, 900)):
            c = synth.make_quiva(n, seed=seed, mean=mean)
            assert ctx.dexqv(c.text) == O.dexqv(c.text)
        c = synth.make_quiva(120, seed=9, mean=5000)                    # bytes >= 128: those entries go by the text-reading kernels
        txt = bytearray(c.text)
        for i in (3, 50, 51, 119):
            L_, o = int(c.len[i]), int(c.off[i])
            txt[o + 4 * (L_ + 1) + 7] = 200                            # substitution line
        got = ctx.dexqv(bytes(txt))
        assert got == O.dexqv(bytes(txt))
        info = ctx.qv_onepass_info()
        assert info["direct"] == 2 and info["text_entries"] == 4
        for case in O.cases("quiva"):                                   # the reference's own bytes
            assert ctx.dexqv(O.golden(case["input"] + ".quiva"), "-l" in case["flags"]) == O.golden(case["name"] + ".dexqv")


_POISON_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import _oracle as O
from dextractor_amd import api, synth
with api.Context(0) as ctx:
    # growing inputs: every internal buffer is re-allocated (and poisoned) again and again, the scratch among them
    for n, mean, seed in [(3, 300, 1), (40, 8000, 3), (700, 900, 4), (24, 30000, 6), (2000, 1500, 9), (5, 200, 2)]:
        c = synth.make_quiva(n, seed=seed, mean=mean)
        for lossy in (0, 1):
            dx = ctx.dexqv(c.text, lossy)
            assert dx == O.dexqv(c.text, lossy), ("dexqv", n, mean, lossy)
        assert ctx.undexqv(ctx.dexqv(c.text), upper=True) == c.text, ("undexqv", n, mean)
    for kind, enc, dec in (("fasta", ctx.dexta, lambda x: ctx.undexta(x, upper=True)), ("arrow", ctx.dexar, ctx.undexar)):
        for n, mean in [(5, 300), (300, 9000), (30, 40000)]:
            f = synth.make_seqfile(kind, n, seed=n, mean=mean)
            img = enc(f.text)
            assert img == (O.dexta(f.text) if kind == "fasta" else O.dexar(f.text)), (kind, n)
            if kind == "fasta":
                assert dec(img) == f.text, (kind, n)
print("POISON_OK")
"""


@pytest.mark.parametrize("poison", ["0xa5", "0xff"])
def test_poisoned_allocations_change_nothing(poison):
    """DEXGPU_POISON fills every device allocation of the library with a byte: a kernel that reads memory nothing has
    written (fresh device memory usually reads as zeros and hides it) then fails here instead of in the field.  Found
    this way: dx_qv_encode_onepass took a re-allocated scratch buffer that came back at its old address for the old
    buffer and kept a slot layout that was gone (a GPU memory fault when the pages held another process's leftovers)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _POISON_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DEXGPU_POISON=poison), capture_output=True, timeout=600)
    assert r.returncode == 0 and b"POISON_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("mode", ["1", "2"])
def test_follow_route_writes_the_same_stream(monkeypatch, mode):
    """DEXGPU_FOLLOW: the encoder moves every record from its scratch slot to its place itself, one entry behind its
    coding (follow_copy, dx_qv_fast.hpp) -- no compaction kernel.  Same bytes, same index beside them, as the route with
    the compaction kernel: ragged lengths, several groups in one region (a scratch budget), entries that come by the
    text-reading kernel, a d_out that is too small."""
    with api.Context(0) as ctx:
        c = Corpus(ctx, n=60_000, mean=6000)
        ref, info0, _ = c.encode(ctx, 0)
        rec_ref = c.d_rec.download(np.uint64, c.n + 1)
        seg_ref = c.d_seg.download(np.uint32, 5 * c.n)
        monkeypatch.setenv("DEXGPU_FOLLOW", mode)           # 1: the encoder's waves place the records, 2: k_qv_follow beside the encoder
        for budget in (0, 600 << 20 if mode == "1" else 1200 << 20):
            got, info, _ = c.encode(ctx, budget)
            assert info["direct"] == 3 and (info["groups"] == 1 if budget == 0 else info["groups"] > 1), info
            assert len(got) == len(ref) and (got == ref).all()
            assert (c.d_rec.download(np.uint64, c.n + 1) == rec_ref).all()
            assert (c.d_seg.download(np.uint32, 5 * c.n) == seg_ref).all()
        ctx.set_scratch_budget(0)
        p = ctx.qv_prescan(c.batch)
        hist, tot = ctx.qv_hist(c.batch, p)
        ctx.qv_set_coding(api.qv_build(hist, tot, p, False), False)
        cap = len(ref) // 2
        d_out = ctx.to_device(np.full(cap + 4096, 0xEE, np.uint8))
        with pytest.raises(L.DexGPUError) as e:
            ctx.qv_encode_onepass(c.batch, c.d_hdr, c.d_hoff, c.d_seg, c.d_rec, d_out, cap)
        assert e.value.code == -8
        assert (d_out.download(np.uint8, 4096, offset=cap) == 0xEE).all()


@pytest.mark.parametrize("mode", ["1", "2"])
def test_follow_route_small_files_and_text_entries(monkeypatch, mode):
    monkeypatch.setenv("DEXGPU_FOLLOW", mode)
    with api.Context(0) as ctx:
        for seed, n, mean in ((3, 1, 50), (4, 7, 3000), (5, 300, 900), (6, 65, 100), (7, 129, 2000)):
            c = synth.make_quiva(n, seed=seed, mean=mean)
            assert ctx.dexqv(c.text) == O.dexqv(c.text)
        c = synth.make_quiva(120, seed=9, mean=5000)                    # bytes >= 128: those entries go by the text-reading kernel
        txt = bytearray(c.text)
        for i in (3, 50, 51, 119):
            L_, o = int(c.len[i]), int(c.off[i])
            txt[o + 4 * (L_ + 1) + 7] = 200                            # substitution line
        got = ctx.dexqv(bytes(txt))
        assert got == O.dexqv(bytes(txt))
        info = ctx.qv_onepass_info()
        assert info["direct"] == 3 and info["text_entries"] == 4
        for case in O.cases("quiva"):                                   # the reference's own bytes
            assert ctx.dexqv(O.golden(case["input"] + ".quiva"), "-l" in case["flags"]) == O.golden(case["name"] + ".dexqv")


def test_hybrid_route_writes_the_same_stream(monkeypatch):
    """DEXGPU_HYBRID: the last group goes the direct way (sizes on the side stream beside the first group's encode, records
    written in place beside its compaction).  Same bytes and the same index beside them, also in more than two groups."""
    with api.Context(0) as ctx:
        c = Corpus(ctx, n=60_000, mean=6000)
        ref, info0, _ = c.encode(ctx, 0)
        rec_ref = c.d_rec.download(np.uint64, c.n + 1)
        seg_ref = c.d_seg.download(np.uint32, 5 * c.n)
        monkeypatch.setenv("DEXGPU_HYBRID", "1")
        for groups in ("2", "5"):
            monkeypatch.setenv("DEXGPU_ONEPASS_GROUPS", groups)
            got, info, _ = c.encode(ctx, 0)
            assert info["groups"] == int(groups)
            assert len(got) == len(ref) and (got == ref).all()
            assert (c.d_rec.download(np.uint64, c.n + 1) == rec_ref).all()
            assert (c.d_seg.download(np.uint32, 5 * c.n) == seg_ref).all()
