"""The routes dx_qv_encode_onepass takes under memory pressure, at a size where they matter (needs an MI355X).

BASELINE configs[4] gives every GPU a 2.5 M-entry slice (125 GB of QV bytes): beside it, its output and its tokens
there is room for only a fraction of the scratch a 1 M-entry batch gets, so the encoder works in more, smaller
groups or, with no room at all, sizes first and in place.  Here the same pressure is put on a batch that fits a test
(dx_set_scratch_budget scales the memory the encoder may assume down by the same factor): every route must write the
same bytes, and those bytes are pinned to the oracle on a sample."""
import ctypes as C

import numpy as np
import pytest

from _flags import set_flag, test_env

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


def test_onepass_routes_write_the_same_stream(monkeypatch):
    """The product route (sizes from the entries' own histograms, records in place: direct == 2), the one whose sizes come from
    tokens and plain lines (DEXGPU_TEST=sizes_from_tokens: direct == 1; what runs when a batch has one run character only) and
    the one without tokens (no_tokens: sizes and records from the text, direct == 4): all the same stream, and the oracle's."""
    with api.Context(0) as ctx:
        c = Corpus(ctx)
        prod, info_p, coding = c.encode(ctx, 0)
        assert info_p["direct"] == 2 and info_p["groups"] == 0 and info_p["tokens"] == 1 and info_p["text_entries"] == 0
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
            assert prod[int(rec[i]) + hl: int(rec[i + 1])].tobytes() == want, i
        set_flag(monkeypatch, "sizes_from_tokens", "1")
        got, info, _ = c.encode(ctx, 0)
        assert info["direct"] == 1 and info["tokens"] == 1 and len(got) == len(prod) and (got == prod).all()
        set_flag(monkeypatch, "sizes_from_tokens", None)
        set_flag(monkeypatch, "no_tokens", "1")
        got, info, _ = c.encode(ctx, 0)
        assert info["direct"] == 4 and info["tokens"] == 0 and info["text_entries"] == c.n
        assert len(got) == len(prod) and (got == prod).all()


def test_a_scratch_budget_shapes_nothing_any_more(monkeypatch):
    """Records are written in place: there are no scratch regions for a budget (dx_set_scratch_budget, DEXGPU_SCRATCH_BUDGET) to
    shape -- the same route, the same bytes; the file driver under the same budget."""
    with api.Context(0) as ctx:
        c = Corpus(ctx, n=3000, mean=4000)
        a, info_a, _ = c.encode(ctx, 0)
        monkeypatch.setenv("DEXGPU_SCRATCH_BUDGET", str(16 << 20))
        b, info_b, _ = c.encode(ctx, 0)
        assert info_b["direct"] == 2 and (a == b).all()
        b, info_b, _ = c.encode(ctx, 64 << 20)
        assert info_b["direct"] == 2 and (a == b).all()
        ctx.set_scratch_budget(0)
        small = synth.make_quiva(40, seed=5, mean=3000)
        assert ctx.dexqv(small.text) == O.dexqv(small.text)


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
    """DEXGPU_TEST=poison=<byte> fills every device allocation of the library with a byte: a kernel that reads memory nothing has
    written (fresh device memory usually reads as zeros and hides it) then fails here instead of in the field.  Found
    this way: dx_qv_encode_onepass took a re-allocated scratch buffer that came back at its old address for the old
    buffer and kept a slot layout that was gone (a GPU memory fault when the pages held another process's leftovers)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _POISON_SCRIPT.format(root=root, tests=os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DEXGPU_TEST=test_env(poison=poison)), capture_output=True, timeout=600)
    assert r.returncode == 0 and b"POISON_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_file_larger_than_the_text_budget_goes_in_slices(monkeypatch, tmp_path):
    """dx_file_dexqv / dx_file_dexqv_to on a .quiva image 1.5 to 6 times the text the device may hold at once (DEXGPU_TEXT_BUDGET):
    the scan pass slice by slice (histograms summed on the host, delChar / subChar and their first entries carried along), the
    tables, then every slice again -- upload, tokens, encode, records out -- under a scratch budget as well.  The bytes are
    those of the one-shot driver and of the oracle; through the CLI too (the sink writes the slices at their offsets)."""
    import os, subprocess
    with api.Context(0) as ctx:
        c = synth.make_quiva(700, seed=77, mean=9000, dist="lognormal")           # ~32 MB of text, entries of 2 .. 40 kb
        whole = ctx.dexqv(c.text)
        assert whole == O.dexqv(c.text)
        for budget in (len(c.text) * 2 // 3, 5 << 20):                            # 2 slices; 7 slices of 5 MiB
            monkeypatch.setenv("DEXGPU_TEXT_BUDGET", str(budget))
            ctx.set_scratch_budget(64 << 20)
            assert ctx.dexqv(c.text) == whole
            assert ctx.dexqv(c.text, True) == O.dexqv(c.text, True)
            ctx.set_scratch_budget(0)
        late = synth.make_quiva(300, seed=78, mean=9000)                          # the first 'N' tag far into the file: delChar found
        t = bytearray(late.text)                                                 # in a later slice than the first
        for i in range(200):
            L_, o = int(late.len[i]), int(late.off[i])
            t[o + L_ + 1: o + 2 * L_ + 1] = bytes(t[o + L_ + 1: o + 2 * L_ + 1]).replace(b"N", b"A").replace(b"n", b"a")
        t = bytes(t)
        monkeypatch.setenv("DEXGPU_TEXT_BUDGET", str(5 << 20))
        assert ctx.dexqv(t) == O.dexqv(t)
    src = tmp_path / "big.quiva"
    src.write_bytes(c.text)
    tool = os.path.join(os.path.dirname(L.LIB_PATH), "bin", "dexqv")
    if os.path.isfile(tool):
        r = subprocess.run([tool, "-k", str(src)], env=dict(os.environ, DEXGPU_TEXT_BUDGET=str(5 << 20)), capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        assert (tmp_path / "big.dexqv").read_bytes() == whole


def test_text_larger_than_the_budget_comes_out_in_slices(monkeypatch, tmp_path):
    """dx_file_undexqv on a .dexqv whose text is 1.5 to 30 times what the device may hold at once (DEXGPU_TEXT_BUDGET): slices of
    whole entries, decoded into one buffer that goes out before the next slice comes in -- with the image resident (uploaded
    once, or there already after a plan made on the device) and with the slices' bytes uploaded one by one
    (DEXGPU_TEST=slice_input).  The oracle's text; through the CLI too."""
    import os, subprocess
    with api.Context(0) as ctx:
        c = synth.make_quiva(500, seed=91, mean=7000, dist="lognormal")            # ~17 MB of text
        img = O.dexqv(c.text)
        want = O.undexqv(img)
        for budget in (len(want) * 2 // 3, 1 << 20, 65536):
            monkeypatch.setenv("DEXGPU_TEXT_BUDGET", str(budget))
            assert ctx.undexqv(img) == want
            assert ctx.undexqv(img, upper=True) == O.undexqv(img, upper=True)
            set_flag(monkeypatch, "slice_input", "1")
            assert ctx.undexqv(img) == want
            set_flag(monkeypatch, "slice_input", None)
            set_flag(monkeypatch, "device_walk_min", "0")                      # the plan on the device: image and index stay there
            set_flag(monkeypatch, "walk_piece", "8192")
            assert ctx.undexqv(img) == want
            set_flag(monkeypatch, "device_walk_min", None); set_flag(monkeypatch, "walk_piece", None)
    src = tmp_path / "big.dexqv"
    src.write_bytes(img)
    tool = os.path.join(os.path.dirname(L.LIB_PATH), "bin", "undexqv")
    if os.path.isfile(tool):
        r = subprocess.run([tool, "-k", str(src)], env=dict(os.environ, DEXGPU_TEXT_BUDGET=str(1 << 20)), capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        assert (tmp_path / "big.quiva").read_bytes() == want


def test_two_bit_drivers_in_slices(monkeypatch, tmp_path):
    """dx_file_pack2 / dx_file_unpack2[_to] on files 1.5 to 100 times what the device may hold at once (DEXGPU_TEXT_BUDGET): slices
    of whole reads in both directions (dexta.c:104-205, undexta.c:175-271 go read by read).  The oracle's bytes; through the
    CLI (the streaming sink) too."""
    import os, subprocess
    with api.Context(0) as ctx:
        for kind in ("fasta", "arrow"):
            c = synth.make_seqfile(kind, 900, seed=5, mean=6000, width=70)
            want = O.dexta(c.text) if kind == "fasta" else O.dexar(c.text)
            back = O.undexta(want) if kind == "fasta" else O.undexar(want)
            for budget in (len(c.text) * 2 // 3, 1 << 20, 65536):
                monkeypatch.setenv("DEXGPU_TEXT_BUDGET", str(budget))
                assert (ctx.dexta(c.text) if kind == "fasta" else ctx.dexar(c.text)) == want
                assert (ctx.undexta(want) if kind == "fasta" else ctx.undexar(want)) == back
                if kind == "fasta":
                    assert ctx.undexta(want, upper=True, width=33) == O.undexta(want, upper=True, width=33)
                    got = bytearray(len(back))
                    def sink(data, at):
                        got[at: at + len(data)] = data
                    assert ctx.unpack2_stream(want, sink) == len(back) and bytes(got) == back
            monkeypatch.delenv("DEXGPU_TEXT_BUDGET")
    src = tmp_path / "reads.fasta"
    src.write_bytes(c.text if kind == "fasta" else synth.make_seqfile("fasta", 900, seed=5, mean=6000, width=70).text)
    bindir = os.path.join(os.path.dirname(L.LIB_PATH), "bin")
    if os.path.isfile(os.path.join(bindir, "dexta")):
        env = dict(os.environ, DEXGPU_TEXT_BUDGET=str(1 << 20))
        text = src.read_bytes()
        r = subprocess.run([os.path.join(bindir, "dexta"), "-k", str(src)], env=env, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        assert (tmp_path / "reads.dexta").read_bytes() == O.dexta(text)
        os.remove(src)
        r = subprocess.run([os.path.join(bindir, "undexta"), "-k", str(tmp_path / "reads.dexta")], env=env, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        assert (tmp_path / "reads.fasta").read_bytes() == O.undexta(O.dexta(text))
