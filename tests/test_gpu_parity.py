"""Parity of the HIP path against the oracle and the reference's golden bytes (needs an MI355X).

Everything goes through the C-ABI of libdexgpu.so (ctypes).  Bar: bit-exact."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from _flags import set_flag, test_env

import _oracle as O
from dextractor_amd import _lib as L
from dextractor_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def _undex_args(flags):
    upper = "-U" in flags
    width = 80
    for f in flags:
        if f.startswith("-w"):
            width = int(f[2:])
    return upper, width


# ---- golden fixtures: outputs of the compiled reference ----------------------------------------

@pytest.mark.parametrize("case", O.cases("fasta"), ids=lambda c: c["name"])
def test_dexta_golden(ctx, case):
    txt, dx = O.golden(case["name"] + ".fasta"), O.golden(case["name"] + ".dexta")
    assert ctx.dexta(txt) == dx
    upper, width = _undex_args(case["undex_flags"])
    rt = txt if case["rt_is_input"] else O.golden(case["name"] + ".rt.fasta")
    assert ctx.undexta(dx, upper, width) == rt


@pytest.mark.parametrize("case", O.cases("arrow"), ids=lambda c: c["name"])
def test_dexar_golden(ctx, case):
    txt, dx = O.golden(case["name"] + ".arrow"), O.golden(case["name"] + ".dexar")
    assert ctx.dexar(txt) == dx
    _, width = _undex_args(case["undex_flags"])
    rt = txt if case["rt_is_input"] else O.golden(case["name"] + ".rt.arrow")
    assert ctx.undexar(dx, width) == rt


@pytest.mark.parametrize("tokens", [True, False], ids=["tokens", "text"])
@pytest.mark.parametrize("case", O.cases("quiva"), ids=lambda c: c["name"])
def test_dexqv_golden(ctx, case, tokens, monkeypatch):
    """The reference's own .dexqv bytes through the file driver: with the token hand-over between the scan and
    the encoder (k_qv_encode_fast) and, DEXGPU_TEST=no_tokens set, with the generic kernel reading the text."""
    if not tokens:
        set_flag(monkeypatch, "no_tokens", "1")
    txt, dx = O.golden(case["input"] + ".quiva"), O.golden(case["name"] + ".dexqv")
    got = ctx.dexqv(txt, "-l" in case["flags"])
    assert len(got) == len(dx)
    assert got == dx


def test_config1_hash(ctx):
    h = O.hashes()["config1_fasta"]
    c = synth.make_seqfile("fasta", h["n"], seed=h["seed"], mean=h["mean"])
    dx = ctx.dexta(c.text)
    assert hashlib.sha256(dx).hexdigest() == h["dexta_sha256"]
    assert ctx.undexta(dx, upper=True) == c.text                      # round trip byte-identical


@pytest.mark.parametrize("lossy", [0, 1])
def test_config4_sample_hash(ctx, lossy):
    h = O.hashes()["config4s_quiva" + ("_lossy" if lossy else "")]
    c = synth.make_quiva(h["n"], seed=h["seed"], mean=h["mean"])
    dx = ctx.dexqv(c.text, lossy)
    assert len(dx) == h["dexqv_bytes"]
    assert hashlib.sha256(dx).hexdigest() == h["dexqv_sha256"]


# ---- differential against the oracle on seeded corpora -----------------------------------------

@pytest.mark.parametrize("kind", ["fasta", "arrow"])
@pytest.mark.parametrize("seed,n,mean,width", [(1, 40, 50, 80), (2, 300, 1500, 80), (3, 64, 5000, 61), (4, 5, 70000, 80)])
def test_pack2_vs_oracle(ctx, kind, seed, n, mean, width):
    c = synth.make_seqfile(kind, n, seed=seed, mean=mean, width=width)
    if kind == "fasta":
        want = O.dexta(c.text)
        assert ctx.dexta(c.text) == want
        for upper, w in ((True, width), (False, 1), (False, 7), (True, 100000)):
            assert ctx.undexta(want, upper, w) == O.undexta(want, upper, w)
    else:
        want = O.dexar(c.text)
        assert ctx.dexar(c.text) == want
        for w in (width, 3, 1024):
            assert ctx.undexar(want, w) == O.undexar(want, w)


def test_pack2_every_small_length(ctx):
    """Reads of every length 0..200 and around the 1 KiB step / 16-byte lane boundaries."""
    lens = np.array(list(range(0, 200)) + [1008, 1023, 1024, 1025, 1039, 1040, 2047, 2048, 2049, 4096 + 17],
                    dtype=np.uint32)
    for kind in ("fasta", "arrow"):
        c = synth.make_seqfile(kind, len(lens), seed=11, lens=lens)
        want = O.dexta(c.text) if kind == "fasta" else O.dexar(c.text)
        got = ctx.dexta(c.text) if kind == "fasta" else ctx.dexar(c.text)
        assert got == want
        if kind == "fasta":
            assert ctx.undexta(want, True, 80) == c.text
        else:
            assert ctx.undexar(want, 80) == c.text


@pytest.mark.parametrize("width", [1, 2, 14, 15, 16, 17, 18, 31, 63, 64, 65, 127, 1023, 1024, 1025, 4000])
def test_unpack2_line_widths(ctx, width):
    """undexta/undexar -w: widths around the 16-byte lane (several line ends per lane below 16, the
    one-line-end fast path from 16 up) and around the 1 KiB step; reads whose last line is full,
    one short, and empty reads (undexta.c:263-270)."""
    lens = np.array([0, 1, 15, 16, 17, 33, width, width + 1, 2 * width, 2 * width - 1, 3 * width + 5, 1024, 5000, 12345],
                    dtype=np.uint32)
    c = synth.make_seqfile("fasta", len(lens), seed=23, lens=lens)
    img = O.dexta(c.text)
    assert ctx.undexta(img, False, width) == O.undexta(img, False, width)
    c = synth.make_seqfile("arrow", len(lens), seed=24, lens=lens)
    img = O.dexar(c.text)
    assert ctx.undexar(img, width) == O.undexar(img, width)


def test_pack2_arbitrary_bytes(ctx):
    """Every byte value as a 'base': the maps of DB.c:393-441 (bytes >= 128 -> 0 / 3)."""
    body = bytes(b for b in range(256) if b not in (10, 62)) * 3
    fa = b">mv/1/0_%d RQ=0.8\n" % len(body) + body + b"\n"
    assert ctx.dexta(fa)[-(len(body) + 3) // 4:] == O.dexta(fa)[-(len(body) + 3) // 4:]
    ar = b">mv/1/0_%d SN=1.00,2.00,3.00,4.00\n" % len(body) + body + b"\n"
    assert ctx.dexar(ar) == O.dexar(ar)


def _upload_quiva(ctx, c):
    d_text = ctx.to_device(np.frombuffer(c.text, np.uint8))
    d_off, d_len = ctx.to_device(c.off), ctx.to_device(c.len)
    return ctx.qv_batch(d_text, d_off, d_len, len(c.len), text_bytes=len(c.text)), (d_text, d_off, d_len)


@pytest.mark.parametrize("seed,n,mean", [(1, 3, 300), (2, 13, 9000), (3, 40, 8000), (4, 700, 900), (5, 3000, 120)])
def test_qv_scan_vs_oracle(ctx, seed, n, mean):
    """k_qv_prescan + k_qv_hist == QVcoding_Scan (order-dependent start points included)."""
    c = synth.make_quiva(n, seed=seed, mean=mean)
    st = O.qv_scan(c.text)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    assert (p.delChar, p.subChar) == (st.delChar, st.subChar)
    assert (p.del_first, p.sub_first) == (st.del_first, st.sub_first)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert tot == st.totChar
    for s in range(6):
        assert (hist[s] == want[s]).all(), f"histogram {s}"


def test_qv_scan_in_one_call_like_the_two(monkeypatch):
    """dx_qv_scan == dx_qv_prescan + dx_qv_hist whatever the context guessed from its last batch: the guess holding (the same
    kind of batch again), missing (another run density: the other instance of k_qv_hist), the token buffers too small (a
    larger batch), no run character at all, a start state handed in -- and the encode that follows gives the oracle's bytes."""
    cases = [("first", synth.make_quiva(150, seed=31, mean=3000)),
             ("again", synth.make_quiva(150, seed=32, mean=3000)),
             ("dense tokens", synth.make_quiva(150, seed=33, mean=3000, prof=synth.pacbio_profile(0.4, 0.35))),
             ("sparse again", synth.make_quiva(140, seed=34, mean=3000)),
             ("larger", synth.make_quiva(420, seed=35, mean=4000)),
             ("smaller", synth.make_quiva(30, seed=36, mean=800))]
    nodel = synth.make_quiva(60, seed=37, mean=2000)
    t = bytearray(nodel.text)                                  # no 'N' tag anywhere: no delChar (QV.c:993-1002)
    for i in range(60):
        L_, o = int(nodel.len[i]), int(nodel.off[i])
        t[o + L_ + 1: o + 2 * L_ + 1] = bytes(t[o + L_ + 1: o + 2 * L_ + 1]).replace(b"N", b"A").replace(b"n", b"a")
    nodel.text = bytes(t)
    cases.insert(4, ("no deletion run character", nodel))
    with api.Context(0) as ctx:
        for what, c in cases:
            st = O.qv_scan(c.text)
            b, keep = _upload_quiva(ctx, c)
            p, hist, tot = ctx.qv_scan(b)
            want = O.hist_array(st)
            want[4:6] -= 1
            assert (p.delChar, p.subChar, p.del_first, p.sub_first) == (st.delChar, st.subChar, st.del_first, st.sub_first), what
            assert tot == st.totChar and (hist == want).all(), what
            p2 = ctx.qv_prescan(b)                             # ... and the two calls, in the same context
            h2, t2 = ctx.qv_hist(b, p2)
            assert (p2.delChar, p2.subChar, p2.del_first, p2.sub_first) == (p.delChar, p.subChar, p.del_first, p.sub_first), what
            assert t2 == tot and (h2 == hist).all(), what
            assert ctx.dexqv(c.text) == O.dexqv(c.text), what
        # a start state handed in (a later slice of a file): kept, as dx_qv_prescan keeps it
        c = cases[1][1]
        b, keep = _upload_quiva(ctx, c)
        ctx.qv_scan(b)                                         # (something to guess from)
        given = L.QVParams(ord("3"), ord("@"), 5, 7)
        p, hist, tot = ctx.qv_scan(b, entry0=100, params=L.QVParams(ord("3"), ord("@"), 5, 7))
        h2, t2 = ctx.qv_hist(b, given, entry0=100)
        assert (p.delChar, p.subChar, p.del_first, p.sub_first) == (ord("3"), ord("@"), 5, 7)
        assert t2 == tot and (h2 == hist).all()
        set_flag(monkeypatch, "no_scan_guess", "1")        # the switch: the two calls
        p, hist, tot = ctx.qv_scan(b)
        st = O.qv_scan(c.text)
        assert (p.delChar, p.subChar) == (st.delChar, st.subChar) and tot == st.totChar


def test_dense_stretches_in_a_sparse_batch(ctx):
    """A batch whose sampled run density picks the histogram kernel with the short token list (512 tokens a step), with
    stretches in which nearly every symbol is a token: those entries' tokens are given up, their bytes are the oracle's."""
    rng = np.random.Generator(np.random.PCG64(23))
    c = synth.make_quiva(90, seed=41, mean=6000)
    st0 = O.qv_scan(c.text)
    txt = bytearray(c.text)
    for i in range(90):
        L_, o = int(c.len[i]), int(c.off[i])
        if i % 9 != 2 or L_ < 4000:
            continue
        a0 = int(rng.integers(0, L_ - 3000))
        n_ = int(rng.integers(700, 2900))                          # one to three steps' worth
        for line, rc in ((0, st0.delChar), (4, st0.subChar)):
            vals = rng.integers(40, 60, n_).astype(np.uint8)
            vals[vals == rc] = 61
            txt[o + line * (L_ + 1) + a0: o + line * (L_ + 1) + a0 + n_] = vals.tobytes()
        txt[o + (L_ + 1) + a0: o + (L_ + 1) + a0 + n_] = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n_))
    c.text = bytes(txt)
    st = O.qv_scan(c.text)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert (p.delChar, p.subChar) == (st.delChar, st.subChar) and tot == st.totChar and (hist == want).all()
    assert ctx.dexqv(c.text) == O.dexqv(c.text)
    assert ctx.undexqv(ctx.dexqv(c.text), upper=True).split(b"\n")[1::6] == c.text.split(b"\n")[1::6]


def test_an_entry_of_a_million_symbols_with_one_insertion_value(ctx):
    """An entry of 1.3 million symbols whose insertion and merge lines hold one value each: the wave's own counters (eight and four
    copies a symbol, 64 additions a copy and step at most) take all of it, its 16-bit counters in memory are not used (the entry is
    the text-reading kernels'), the workgroup's tables get the full counts.  Histogram and file as the oracle's."""
    lens = np.array([1_300_000, 700, 70_000, 5], np.uint32)
    c = synth.make_quiva(len(lens), seed=77, lens=lens)
    txt = bytearray(c.text)
    L, o = int(c.len[0]), int(c.off[0])
    txt[o + 2 * (L + 1): o + 2 * (L + 1) + L] = b"7" * L            # the insertion line: one value
    txt[o + 3 * (L + 1): o + 3 * (L + 1) + L] = b"9" * L            # ... and the merge line
    c.text = bytes(txt)
    st = O.qv_scan(c.text)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert tot == st.totChar and (hist == want).all()
    assert hist[1, ord("7")] >= L
    assert ctx.dexqv(c.text) == O.dexqv(c.text)


def test_qv_scan_wide_bytes_and_long_runs(ctx):
    """Bytes >= 128 and runs >= 64 leave the conflict-free 32-copy bins of k_qv_hist for its plain
    tables; the file still encodes byte-identically."""
    rng = np.random.Generator(np.random.PCG64(17))
    c = synth.make_quiva(60, seed=9, mean=5000)
    st0 = O.qv_scan(c.text)
    txt = bytearray(c.text)
    for i in range(60):
        L, o = int(c.len[i]), int(c.off[i])
        for line in (0, 2, 3, 4):
            at = o + line * (L + 1)
            for pos in rng.integers(0, L, max(1, L // 40)):
                v = int(rng.integers(128, 256))
                if line != 0 or txt[o + (L + 1) + int(pos)] not in b"nN":       # keep the N-tag <-> delChar pairing
                    txt[at + int(pos)] = v
        if i % 7 == 3 and L > 700:                                # long runs of both run characters
            txt[o + 100: o + 600] = bytes([st0.delChar]) * 500
            txt[o + (L + 1) + 100: o + (L + 1) + 600] = b"N" * 500
            if st0.subChar >= 0:
                txt[o + 4 * (L + 1) + 50: o + 4 * (L + 1) + 450] = bytes([st0.subChar]) * 400
    c.text = bytes(txt)
    st = O.qv_scan(c.text)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    assert (p.delChar, p.subChar, p.del_first, p.sub_first) == (st.delChar, st.subChar, st.del_first, st.sub_first)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert tot == st.totChar and (hist == want).all()
    assert hist[:4, 128:].sum() > 1000 and hist[4:, 64:].sum() > 10
    assert ctx.dexqv(c.text) == O.dexqv(c.text)


def test_qv_scan_late_delchar_and_none(ctx):
    txt = O.golden("qv_runs.quiva")
    off, ln, hdr, _ = api.index_quiva(txt)
    st = O.qv_scan(txt)
    assert st.del_first == 2
    c = synth.Corpus(txt, off, ln, hdr)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    assert (p.delChar, p.del_first, p.subChar) == (st.delChar, 2, st.subChar)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st); want[4:6] -= 1
    assert (hist == want).all()
    txt = O.golden("qv_nodel.quiva")
    off, ln, hdr, _ = api.index_quiva(txt)
    b, keep = _upload_quiva(ctx, synth.Corpus(txt, off, ln, hdr))
    p = ctx.qv_prescan(b)
    assert p.delChar == -1 and p.del_first == -1


@pytest.mark.parametrize("lossy", [0, 1])
@pytest.mark.parametrize("seed,n,mean", [(1, 3, 300), (2, 13, 9000), (3, 40, 8000), (4, 700, 900), (6, 24, 30000)])
def test_dexqv_vs_oracle(ctx, seed, n, mean, lossy):
    c = synth.make_quiva(n, seed=seed, mean=mean)
    assert ctx.dexqv(c.text, lossy) == O.dexqv(c.text, lossy)


@pytest.mark.parametrize("run_p", [0.02, 0.3, 0.6, 0.97, 0.999])
def test_dexqv_run_densities(ctx, run_p):
    prof = synth.pacbio_profile(del_run_p=run_p, sub_run_p=run_p)
    c = synth.make_quiva(30, seed=77, mean=9000, prof=prof)
    assert ctx.dexqv(c.text) == O.dexqv(c.text)


def test_dexqv_every_small_length(ctx):
    """Entries of every length 1..130 and around the step/lane boundaries; sizes and segments."""
    lens = np.array(list(range(1, 131)) + [1007, 1008, 1009, 1023, 1024, 1025, 1040, 2048, 2049, 3000, 5000,
                                           16384, 16385] + [9000] * 24, dtype=np.uint32)
    c = synth.make_quiva(len(lens), seed=31, lens=lens)
    want = O.dexqv(c.text)
    assert ctx.dexqv(c.text) == want


def test_qv_sizes_and_segments_vs_oracle(ctx):
    """k_qv_sizes record offsets and the k_qv_encode segment index against per-entry oracle calls."""
    c = synth.make_quiva(60, seed=8, mean=4000)
    st = O.qv_scan(c.text)
    ref_coding = O.qv_create(st)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    coding = api.qv_build(hist, tot, p)
    ctx.qv_set_coding(coding)
    blob, hoff, _ = api.frame_headers(c.hdr)
    d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
    n = len(c.len)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(4 * 5 * n)
    total = ctx.qv_sizes(b, d_hoff, d_seg, d_rec)
    rec = d_rec.download(np.uint64)
    d_out = ctx.alloc(total)
    ctx.qv_encode(b, d_hdr, d_hoff, d_rec, d_seg, d_out)
    out = d_out.download(np.uint8, total).tobytes()
    seg = d_seg.download(np.uint32, 5 * n).reshape(n, 5)
    text = np.frombuffer(c.text, np.uint8)
    at = 0
    for i in range(n):
        L = int(c.len[i]); o = int(c.off[i])
        lines = np.stack([text[o + k * (L + 1): o + k * (L + 1) + L] for k in range(5)])
        body, want_seg = O.qv_encode_entry(ref_coding, False, lines)
        hl = int(hoff[i + 1] - hoff[i])
        assert int(rec[i]) == at
        assert list(seg[i]) == want_seg
        assert out[at + hl: at + hl + len(body)] == body
        at += hl + len(body)
    assert int(rec[n]) == at == total


def test_qv_type2_and_pad_rule(ctx):
    """Fibonacci-weighted symbols (8-bit escapes) in every stream position + all pad-rule branches."""
    rng = np.random.Generator(np.random.PCG64(3))
    f = [1, 1]
    while len(f) < 23:
        f.append(f[-1] + f[-2])
    pool = np.concatenate([np.full(cn, 40 + i, np.uint8) for i, cn in enumerate(f)])
    ents = []
    prof = synth.pacbio_profile()
    for e in range(30):
        L = int(rng.integers(1, 4000))
        b = synth.qv_lines(5, e, L, prof)
        b[2] = rng.choice(pool, L)
        b[3] = rng.choice(pool, L)
        ents.append((10 + e * 7, 0, 800, [b[r].tobytes() for r in range(5)]))
    out = []
    for well, beg, qv, lines in ents:
        L = len(lines[0])
        out.append(b"@m9/%d/%d_%d RQ=0.%d\n" % (well, beg, beg + L, qv))
        out += [x + b"\n" for x in lines]
    txt = b"".join(out)
    want = O.dexqv(txt)
    assert ctx.dexqv(txt) == want


def test_device_generator_matches_numpy(ctx):
    n = 37
    lens = synth.lengths(n, 5, mean=700)
    lens[3] = 1; lens[4] = 16; lens[5] = 1024; lens[6] = 1025
    hdr = synth.headers(n, 5, lens)
    prof = synth.pacbio_profile()
    c = synth.make_quiva(n, seed=5, lens=lens, fixed_width=True, prof=prof)
    d_off, d_len, d_hdr = ctx.to_device(c.off), ctx.to_device(c.len), ctx.to_device(hdr)
    d_lut = ctx.to_device(prof.table())
    d_text = ctx.alloc(len(c.text)).zero()
    ctx.synth_quiva(5, 0, n, d_off, d_len, d_hdr, d_lut, prof.del_run, "m000_000", d_text)
    assert d_text.download(np.uint8, len(c.text)).tobytes() == c.text
    # a slice generated on its own equals the same entries of the whole corpus
    d_text2 = ctx.alloc(len(c.text)).zero()
    ctx.synth_quiva(5, 20, n - 20, ctx.to_device(c.off[20:]), ctx.to_device(c.len[20:]), ctx.to_device(hdr[20:]),
                    d_lut, prof.del_run, "m000_000", d_text2)
    lo = int(c.off[20]) - 43
    assert d_text2.download(np.uint8, len(c.text)).tobytes()[lo:] == c.text[lo:]


def test_errors_are_loud(ctx):
    with pytest.raises(L.DexGPUError) as e:
        ctx.dexqv(b"@m/1/0_3 RQ=0.8\nabc\nabc\nab\nabc\nabc\n")
    assert e.value.code == -3
    with pytest.raises(L.DexGPUError):
        ctx.undexta(b"\x00\x01garbage")
    with pytest.raises(L.DexGPUError):
        ctx.undexta(O.golden("ta_edge.dexta"), False, 0)


# ---- decode (undexqv) --------------------------------------------------------------------------

@pytest.fixture(params=["plain kernel", "generic kernel"])
def decoder(request, monkeypatch):
    """The plain lines are decoded by k_qv_decode_plain (aligned-line input rings, 16 symbols per store); with
    DEXGPU_TEST=generic_decode set everything goes through the generic lane-per-stream kernel: both must agree."""
    if request.param == "generic kernel":
        set_flag(monkeypatch, "generic_decode", "1")
    return request.param


def test_undexqv_ragged_lengths_and_alignments(ctx, decoder):
    """Lengths around the decoder's block (16) and step sizes, entry groups of very unequal lengths (the 64 lanes
    of a wavefront run out at different times), segments at every byte alignment (the well-delta chain and odd
    tag segments shift them), a line ending exactly at a 64-byte line."""
    lens = np.array([0, 1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025,
                     4095, 4096, 4097, 9000, 3, 70001, 5, 12000, 1] + [int(x) for x in np.random.default_rng(5).integers(0, 700, 120)],
                    np.uint32)
    c = synth.make_quiva(len(lens), seed=41, lens=lens)
    dx = O.dexqv(c.text)
    assert ctx.undexqv(dx, upper=True) == c.text
    assert ctx.undexqv(dx, upper=False) == O.undexqv(dx, upper=False)


@pytest.mark.parametrize("case", O.cases("quiva"), ids=lambda c: c["name"])
def test_undexqv_golden(ctx, case, decoder):
    txt, dx = O.golden(case["input"] + ".quiva"), O.golden(case["name"] + ".dexqv")
    rt = txt if case["rt_is_input"] else O.golden(case["name"] + ".rt.quiva")
    assert ctx.undexqv(dx, upper=True) == rt                     # reference `undexqv -U` output
    assert ctx.undexqv(dx, upper=False) == O.undexqv(dx, upper=False)


def test_undexqv_streamed_equals_whole(ctx):
    """dx_file_undexqv_plan + _run (what the CLI uses: the text leaves through a sink in 32 MiB chunks, header lines
    laid over each chunk on its way) against the reference's text -- 90 MB of it, so that header lines straddle
    chunk boundaries -- and a sink that refuses data."""
    c = synth.make_quiva(1800, seed=21, mean=10000)
    dx = ctx.dexqv(c.text)
    got = bytearray(len(c.text))
    chunks = []
    def sink(data, at):
        got[at: at + len(data)] = data
        chunks.append((at, len(data)))
    assert ctx.undexqv_stream(dx, sink, upper=True) == len(c.text)
    assert bytes(got) == c.text
    assert len(chunks) == (len(c.text) + (32 << 20) - 1) // (32 << 20)
    assert [a for a, _ in chunks] == sorted(a for a, _ in chunks) and sum(n for _, n in chunks) == len(c.text)
    with pytest.raises(L.DexGPUError) as e:
        ctx.undexqv_stream(dx, lambda data, at: at > 0, upper=True)
    assert e.value.code == -9
    # dx_file_dexqv_to: the .dexqv image through a sink (head first, then the record stream in chunks)
    parts = {}
    assert ctx.dexqv_stream(c.text, lambda data, at: parts.__setitem__(at, data) and False) == len(dx)
    assert b"".join(parts[k] for k in sorted(parts)) == dx and len(parts) >= 2
    with pytest.raises(L.DexGPUError) as e:                       # a malformed file: the sink sees nothing
        ctx.dexqv_stream(c.text[:-7], lambda data, at: parts.__setitem__("bad", data) and False)
    assert e.value.code == -3 and "bad" not in parts
    # dx_file_unpack2_to: 70 MB of fasta text back out of its .dexta image, header lines across chunk boundaries
    f = synth.make_seqfile("fasta", 7000, seed=22, mean=10000)
    dxa = ctx.dexta(f.text)
    back = bytearray(len(f.text))
    assert ctx.unpack2_stream(dxa, lambda data, at: back.__setitem__(slice(at, at + len(data)), data) and False,
                              mode=L.DX_LETTERS_UPPER, width=80) == len(f.text)
    assert bytes(back) == ctx.undexta(dxa, upper=True) == f.text
    for case in O.cases("quiva"):                                 # and the goldens, small: one chunk
        dxg = O.golden(case["name"] + ".dexqv")
        rt = O.golden(case["input"] + ".quiva") if case["rt_is_input"] else O.golden(case["name"] + ".rt.quiva")
        parts = {}
        ctx.undexqv_stream(dxg, lambda data, at: parts.__setitem__(at, data) and False, upper=True)
        assert b"".join(parts[k] for k in sorted(parts)) == rt
        parts = {}
        ctx.dexqv_stream(O.golden(case["input"] + ".quiva"), lambda data, at: parts.__setitem__(at, data) and False,
                         lossy="-l" in case["flags"])
        assert b"".join(parts[k] for k in sorted(parts)) == dxg


@pytest.mark.parametrize("lossy", [0, 1])
@pytest.mark.parametrize("seed,n,mean", [(1, 3, 300), (3, 40, 8000), (4, 700, 900), (6, 24, 30000)])
def test_undexqv_vs_oracle(ctx, seed, n, mean, lossy, decoder):
    c = synth.make_quiva(n, seed=seed, mean=mean)
    dx = O.dexqv(c.text, lossy)
    got = ctx.undexqv(dx, upper=True)
    assert got == O.undexqv(dx, upper=True)
    if not lossy:
        assert got == c.text                                     # round trip byte-identical


def test_device_round_trip_with_encoder_index(ctx):
    """encode -> decode entirely on the device, the decoder fed by the encoder's own index."""
    lens = np.array(list(range(1, 80)) + [1023, 1024, 1025, 4097] + [7000] * 40, dtype=np.uint32)
    c = synth.make_quiva(len(lens), seed=21, lens=lens)
    n = len(lens)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    coding = api.qv_build(hist, tot, p)
    ctx.qv_set_coding(coding)
    blob, hoff, _ = api.frame_headers(c.hdr)
    d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
    total = ctx.qv_sizes(b, d_hoff, d_seg, d_rec)
    d_out = ctx.alloc(total)
    ctx.qv_encode(b, d_hdr, d_hoff, d_rec, d_seg, d_out)
    d_txt2 = ctx.to_device(np.frombuffer(c.text, np.uint8).copy())
    # wipe the data lines, keep the header lines, then decode in place
    img = np.frombuffer(c.text, np.uint8).copy()
    for i in range(n):
        img[int(c.off[i]): int(c.off[i]) + 5 * (int(c.len[i]) + 1)] = 0
    d_txt2.upload(img)
    ctx.qv_decode(d_out, d_rec, d_hoff, d_seg, keep[2], n, True, d_txt2, keep[1])
    assert d_txt2.download(np.uint8, len(c.text)).tobytes() == c.text


@pytest.mark.parametrize("route", ["scratch", "direct", "text"])
@pytest.mark.parametrize("case", ["ragged", "long_codes", "no_runs", "lossy", "long_entries", "odd_entries", "long_runs",
                                  "dense_runs_90", "dense_runs_97", "dense_runs_99"])
def test_decode_with_the_encoders_group_index(ctx, case, route, monkeypatch):
    """dx_qv_subindex: the one-pass encoder (each of its routes) leaves the code bits of every group of 16 symbols of the
    plain lines; dx_qv_decode of that stream in the same context then runs k_qv_decode_sub, a wavefront per line.  Same text
    as without the index, byte for byte -- whole batch, a contiguous part of it, and after the index has gone stale."""
    if route == "direct":
        set_flag(monkeypatch, "sizes_from_tokens", "1")
    if route == "text":
        set_flag(monkeypatch, "no_tokens", "1")
    lossy = case == "lossy"
    if case == "long_runs":                                       # runs of hundreds and thousands: passes that cover more of the line
        c = synth.make_quiva(30, seed=34, mean=20000)             # than the decoder stages (RUN_STRETCH), 16-bit run literals
        txt, rc_d = bytearray(c.text), O.qv_scan(c.text).delChar
        for e, (at, run) in enumerate([(100, 300), (50, 5000), (7, 9000), (0, 126), (3000, 4000), (10, 70000)]):
            Le = int(c.len[e])
            run = min(run, Le - at - 2)
            o_ = int(c.off[e]) + at
            txt[o_: o_ + run] = bytes([rc_d]) * run
            o1 = int(c.off[e]) + (Le + 1) + at
            txt[o1: o1 + run] = b"N" * run
        for e in (10, 11):                                        # hundreds of runs of 100 (tokens stay usable): a pass of 512
            Le, o_ = int(c.len[e]), int(c.off[e])                 # tokens covers ~50 k positions, far more than RUN_STRETCH
            unit_d, unit_t = bytes([rc_d]) * 100 + b"5", b"N" * 100 + b"A"
            k = (Le - 200) // 101
            txt[o_ + 100: o_ + 100 + 101 * k] = unit_d * k
            txt[o_ + (Le + 1) + 100: o_ + (Le + 1) + 100 + 101 * k] = unit_t * k
        c.text = bytes(txt)
    elif case.startswith("dense_runs"):                           # run characters 90 / 97 / 99 % of the two lines: a pass of 512 tokens covers 5 k,
        rp = int(case[-2:]) / 100.0                               # 17 k, 51 k positions -- through the staging buffer 5120 at a time --, runs of
        c = synth.make_quiva(40, seed=36, mean=30000,             # 127 and more in most lanes (exception tokens: the line keeps its index),
                             prof=synth.pacbio_profile(del_run_p=rp, sub_run_p=rp))   # runs of 255 and more (16-bit literals: code by code)
    elif case == "ragged":                                        # every group shape: 0, < 16, multiples of 16, around a step (1024) and a round
        lens = np.array(list(range(0, 70)) + [255, 256, 257, 1023, 1024, 1025, 1040, 2047, 2048, 2049, 4097, 9999, 10000,
                                              16383, 16384, 16385, 16400] + [7000] * 30, np.uint32)
        c = synth.make_quiva(len(lens), seed=31, lens=lens)
    elif case == "long_entries":                                  # many rounds per line
        lens = np.array([70001, 140000, 33000, 16385, 5], np.uint32)
        c = synth.make_quiva(len(lens), seed=32, lens=lens)
    elif case == "odd_entries":                                   # entries of the unusable list go through the generic encoder:
                                                                  # their run-coded lines have no index (RUN_NONE) and take k_qv_decode
        c = synth.make_quiva(40, seed=9, mean=4000)
        txt, rc_d = bytearray(c.text), O.qv_scan(c.text).delChar
        for e, run in ((3, 126), (5, 127), (9, 300)):
            o_ = int(c.off[e]) + 49
            txt[o_: o_ + run + 2] = b"5" + bytes([rc_d]) * run + b"5"
            o1 = int(c.off[e]) + (int(c.len[e]) + 1) + 49
            txt[o1: o1 + run + 2] = b"A" + b"N" * run + b"A"
        c.text = bytes(txt)
    else:
        c = synth.make_quiva(120, seed=33, mean=6000)
    st = O.qv_scan(c.text)
    n = len(c.len)
    b, keep = _upload_quiva(ctx, c)
    if case == "long_codes":                                      # 16-bit codes: a round of 4 steps outgrows the window
        sub = st.subChar if st.subChar >= 0 else int(np.argmax(O.hist_array(st)[3]))
        coding = _fixed_coding(16, 16, True, st.delChar, sub)
        ctx.qv_hist(b, L.QVParams(coding.delChar, coding.subChar, 0, 0))
    elif case == "no_runs":                                       # all four lines plain
        coding = _fixed_coding(7, 9, False, -1, -1)
        ctx.qv_hist(b, L.QVParams(-1, -1, 0, 0))
    else:
        p = ctx.qv_prescan(b)
        hist, tot = ctx.qv_hist(b, p)
        coding = api.qv_build(hist, tot, p, lossy)
    ctx.qv_set_coding(coding, lossy)
    blob, hoff, _ = api.frame_headers(c.hdr)
    d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
    cap = len(c.text) + 4096 * n + 4096
    d_out = ctx.alloc(cap)

    def decode(first=0, count=None, wipe=True):
        count = n - first if count is None else count
        img = np.frombuffer(c.text, np.uint8).copy()
        for i in range(n):
            img[int(c.off[i]): int(c.off[i]) + 5 * (int(c.len[i]) + 1)] = 0
        d_txt = ctx.to_device(img)
        ctx.qv_decode(d_out, d_rec.offset(8 * first), d_hoff.offset(8 * first), d_seg.offset(20 * first),
                      keep[2].offset(4 * first), count, True, d_txt, keep[1].offset(8 * first))
        return d_txt.download(np.uint8, len(c.text)).tobytes()

    ctx.qv_subindex(False)
    ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
    want = decode()                                               # the lane-per-line kernels
    ctx.qv_subindex(True)
    try:
        total = ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
        ctx.profile(True)
        got = decode()
        ran = ctx.kernel_times()
        ctx.profile(False)
        assert got == want
        if not lossy and case not in ("long_codes", "no_runs"):   # (the hand-made tables do not cover every byte value)
            assert got == c.text
        # k_qv_decode_sub + (k_qv_decode_runs + k_qv_decode for what has no index) + tags: no lane-per-line plain kernel
        assert "k_qv_decode_plain" not in ran and ran["k_qv_decode_sub"][1] == 1
        assert sum(v[1] for k, v in ran.items() if k in L.DECODE_KERNELS) == (2 if case == "no_runs" else 4)
        first, count = n // 3, n // 2                             # a contiguous part of the batch
        part = decode(first, count)
        lo, hi = int(c.off[first]), int(c.off[first + count - 1]) + 5 * (int(c.len[first + count - 1]) + 1)
        assert part[lo:hi] == want[lo:hi]
        ctx.qv_set_coding(coding, lossy)                          # tables set again: the index is dropped, the old kernels decode
        assert decode() == want
    finally:
        ctx.qv_subindex(False)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 9, 13])
def test_indexed_decode_of_small_batches(ctx, n):
    """The wave-per-line decoders draw four entries per ticket and request every entry's whereabouts, header words and
    first round an entry ahead: batches smaller than a ticket, not a multiple of one, with empty entries in between and
    at either end, and every contiguous part of them."""
    lens = np.array([(0, 5000, 17, 0, 2500, 1, 1024, 16, 0, 3333, 15, 4096, 0)[i % 13] if n > 2 else (700, 0)[i % 2]
                     for i in range(n)], np.uint32)
    c = synth.make_quiva(n, seed=50 + n, lens=lens)
    b, keep = _upload_quiva(ctx, c)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    coding = api.qv_build(hist, tot, p, False)
    ctx.qv_set_coding(coding, False)
    blob, hoff, _ = api.frame_headers(c.hdr)
    d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
    cap = len(c.text) + 4096 * n + 4096
    d_out = ctx.alloc(cap)
    ctx.qv_subindex(True)
    try:
        ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, cap)
        for first in range(n):
            for count in range(1, n - first + 1):
                img = np.frombuffer(c.text, np.uint8).copy()
                for i in range(n):
                    img[int(c.off[i]): int(c.off[i]) + 5 * (int(c.len[i]) + 1)] = 0
                d_txt = ctx.to_device(img)
                ctx.profile(True)
                ctx.qv_decode(d_out, d_rec.offset(8 * first), d_hoff.offset(8 * first), d_seg.offset(20 * first),
                              keep[2].offset(4 * first), count, True, d_txt, keep[1].offset(8 * first))
                ran = ctx.kernel_times()
                ctx.profile(False)
                assert ran["k_qv_decode_sub"][1] == 1 and "k_qv_decode_plain" not in ran
                got = d_txt.download(np.uint8, len(c.text)).tobytes()
                lo, hi = int(c.off[first]), int(c.off[first + count - 1]) + 5 * (int(c.len[first + count - 1]) + 1)
                assert got[lo:hi] == c.text[lo:hi], (first, count)
                assert got[:lo] == img[:lo].tobytes() and got[hi:] == img[hi:].tobytes(), (first, count)   # nothing outside the part
    finally:
        ctx.qv_subindex(False)


def test_undexqv_no_delchar_and_type2(ctx, decoder):
    for name in ("qv_nodel", "qv_type2", "qv_runs"):
        dx = O.golden(name + ".dexqv")
        assert ctx.undexqv(dx, upper=False) == O.undexqv(dx, upper=False)


def _quiva(ents, movie=b"m7"):
    out = []
    for well, lines in ents:
        L = len(lines[0])
        out.append(b"@%s/%d/%d_%d RQ=0.%d\n" % (movie, well, 3, 3 + L, 801))
        out += [bytes(x) + b"\n" for x in lines]
    return b"".join(out)


SINGLE_SYMBOL_QUIVA = b"""@m000_000/1/1588_1590 RQ=0.797
&2
CN
%#
"E
??
@m000_000/22/4851_4865 RQ=0.777
$22,22222.%22)
CNNGNNNNNTCNNA
""%!""%#!!1$"!
+"!6219,(17(91
??????????????
@m000_000/54/1742_1747 RQ=0.774
"22)2
ANNAN
""#"#
?J*,.
?????
"""


@pytest.mark.parametrize("index", [False, True], ids=["walk", "walk_indexed"])
def test_a_line_of_one_symbol_is_written_like_the_reference_and_refused_like_it(ctx, tmp_path, index, monkeypatch):
    """A file too short for a substitution run character whose substitution lines hold one value only: the Huffman tree
    of that stream is a single leaf, its code has no bits (QV.c:136-147).  The reference writes the file -- and its own
    undexqv then stops with "Could not read more bits (Decode)", exit 1.  Same bytes here, and a loud refusal instead of
    a text of zeros (found by tools/stress_decode.py)."""
    if index:
        set_flag(monkeypatch, "walk_index", "1")
    dx = ctx.dexqv(SINGLE_SYMBOL_QUIVA)
    assert dx == O.dexqv(SINGLE_SYMBOL_QUIVA)
    if O.have_ref():
        assert dx == O.run_ref("dexqv", [], SINGLE_SYMBOL_QUIVA, ".quiva", ".dexqv", tmp_path)
        with pytest.raises(RuntimeError) as e:
            O.run_ref("undexqv", [], dx, ".dexqv", ".quiva", tmp_path)
        assert "Could not read more bits" in str(e.value)
    with pytest.raises(L.DexGPUError) as e:                          # (the host walk of the records stops at the first such code)
        ctx.undexqv(dx, upper=True)
    assert e.value.code == -3                                        # DX_E_FORMAT
    # the same stream decoded on the device with the encoder's own record boundaries: no walk in front of it, the kernels
    # themselves must notice (they used to return a text with zeros in it)
    lens = np.array([2, 14, 5], np.uint32)
    offs, at = [], 0
    for ln in SINGLE_SYMBOL_QUIVA.split(b"@")[1:]:
        h = ln.index(b"\n") + 2
        offs.append(at + h)
        at += len(ln) + 1
    d_text = ctx.to_device(np.frombuffer(SINGLE_SYMBOL_QUIVA, np.uint8))
    d_off, d_len = ctx.to_device(np.array(offs, np.uint64)), ctx.to_device(lens)
    b = ctx.qv_batch(d_text, d_off, d_len, 3, text_bytes=len(SINGLE_SYMBOL_QUIVA))
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    coding = api.qv_build(hist, tot, p, False)
    ctx.qv_set_coding(coding, False)
    d_hoff = ctx.to_device(np.zeros(4, np.uint64))
    d_hdr = ctx.alloc(16)
    d_rec, d_seg, d_out = ctx.alloc(8 * 4), ctx.alloc(20 * 3), ctx.alloc(1 << 16)
    ctx.qv_subindex(index)
    try:
        ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg, d_rec, d_out, 1 << 16)
        d_back = ctx.to_device(np.zeros(len(SINGLE_SYMBOL_QUIVA), np.uint8))
        with pytest.raises(L.DexGPUError) as e:
            ctx.qv_decode(d_out, d_rec, d_hoff, d_seg, d_len, 3, True, d_back, d_off)
        assert e.value.code == -3 and "in no table" in str(e.value)
    finally:
        ctx.qv_subindex(False)
    c = synth.make_quiva(20, seed=3, mean=500)                      # (the context is as good as before)
    assert ctx.undexqv(ctx.dexqv(c.text), upper=True) == c.text


def test_dexqv_zero_length_entries(ctx):
    """Entries with no symbols at all (five empty lines) between normal ones."""
    prof = synth.pacbio_profile()
    ents = []
    for e, L in enumerate([0, 700, 0, 0, 15, 1024, 0]):
        b = synth.qv_lines(3, e, L, prof)
        ents.append((5 + 3 * e, [b[r].tobytes() for r in range(5)]))
    txt = _quiva(ents)
    want = O.dexqv(txt)
    assert ctx.dexqv(txt) == want
    assert ctx.undexqv(want, upper=True) == O.undexqv(want, upper=True)


def test_dexqv_run_longer_than_16_bits(ctx):
    """A deletion run of 70000 does not fit the 16-bit run literal (QV.c:487): the reference ORs the
    overflowing bit into the preceding code.  The device path reproduces those bytes."""
    prof = synth.pacbio_profile()
    L = 70000 + 50
    b = synth.qv_lines(9, 0, L, prof)
    b[0, :] = ord("2"); b[1, :] = ord("N")
    b[0, 70000:] = np.arange(50) % 16 + 34
    b[1, 70000:] = ord("A")
    b2 = synth.qv_lines(9, 1, 3000, prof)
    txt = _quiva([(4, [b[r].tobytes() for r in range(5)]), (9, [b2[r].tobytes() for r in range(5)])])
    assert ctx.dexqv(txt) == O.dexqv(txt)


@pytest.mark.parametrize("name", ["qv_full", "qv_type2", "qv_runs", "qv_nodel"])
def test_undexqv_byteswapped_file(ctx, name, decoder):
    """GETFLIP path (QV.c:553-568): a .dexqv written on a host of the other endianness."""
    dx = O.golden(name + ".dexqv")
    fl = O.byteswap_dexqv(dx, api.qv_walk(dx))
    assert ctx.undexqv(fl, upper=True) == ctx.undexqv(dx, upper=True) == O.undexqv(fl, upper=True)


def test_undexqv_older_layout(ctx):
    """No 0x55aa key and uint16 beg/end/qv (undexqv.c:104-109, 159-179); expected text = the real reference
    undexqv's output for this image (tests/golden/make_golden.py)."""
    leg = O.golden("qv_tiny.legacy.dexqv")
    assert ctx.undexqv(leg, upper=True) == O.golden("qv_tiny.legacy.rt.quiva")
    assert ctx.undexqv(leg, upper=False) == O.golden("qv_tiny.legacy.rt_lower.quiva")


@pytest.mark.parametrize("name", ["qv_full", "qv_lossy", "qv_type2", "qv_mid"])
def test_dexqv_golden_without_pair_tables(ctx, name, monkeypatch):
    """k_qv_encode_fast codes the insertion and merge lines two symbols per look-up when their coded byte
    values span at most 64 (the usual case, all goldens); DEXGPU_TEST=no_pairs keeps the one-symbol step: same bytes.
    qv_type2's insertion scheme has 8-bit escapes: pairs beyond 24 bits fall back step by step."""
    case = [c for c in O.cases("quiva") if c["name"] == name][0]
    txt, dx = O.golden(case["input"] + ".quiva"), O.golden(name + ".dexqv")
    assert ctx.dexqv(txt, "-l" in case["flags"]) == dx
    set_flag(monkeypatch, "no_pairs", "1")
    assert ctx.dexqv(txt, "-l" in case["flags"]) == dx


def test_file_drivers_two_pass_switch_gives_the_same_bytes(ctx, monkeypatch):
    """DEXGPU_TEST=twopass selects dx_qv_sizes + dx_qv_encode in the file drivers instead of the one-pass
    encoder: same file."""
    c = synth.make_quiva(50, seed=23, mean=5000)
    want = O.dexqv(c.text)
    assert ctx.dexqv(c.text) == want
    set_flag(monkeypatch, "twopass", "1")
    assert ctx.dexqv(c.text) == want


def test_undexta_legacy_and_byteswapped_keys(ctx):
    """undexta accepts 0x33cc (uint16 fields) and byte-swapped files (undexta.c:140-159, 211-240)."""
    import struct
    name = b">mv"
    recs = [(5, 0, 7, 851, b"\x1b\x18"), (300, 3, 5, 7, b"\xb0")]      # ACGTACG, GT
    def build(key, flip, newv):
        e = ">" if flip else "<"
        out = struct.pack(e + "H", key) + struct.pack(e + "i", len(name)) + name
        last = 0
        for well, beg, end, qv, body in recs:
            d = well - last
            out += b"\xff" * (d // 255) + bytes([d % 255]); last = well
            out += struct.pack(e + ("iii" if newv else "HHH"), beg, end, qv) + body
        return out
    imgs = [build(0x55aa, False, True), build(0x55aa, True, True), build(0x33cc, False, False), build(0x33cc, True, False)]
    want = b">mv/5/0_7 RQ=0.851\nacgtacg\n>mv/300/3_5 RQ=0.7\ngt\n"
    for k, img in enumerate(imgs):
        assert O.undexta(img) == want
        assert ctx.undexta(img) == want
        if O.have_ref():                                              # ... and the real undexta says the same
            import tempfile
            with tempfile.TemporaryDirectory() as d:
                assert O.run_ref("undexta", [], img, ".dexta", ".fasta", d) == want, k


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_concatenates_to_whole_file(ctx, world):
    """One .quiva cut into contiguous entry ranges ("ranks"), each shard scanned and encoded on its own
    with the host-summed histograms / merged scan state: shard streams concatenate to the whole file."""
    from dextractor_amd import shard
    c = synth.make_quiva(37, seed=61, mean=8000)
    want = ctx.dexqv(c.text)
    assert want == O.dexqv(c.text)
    n = len(c.len)
    text = np.frombuffer(c.text, np.uint8)
    d_text = ctx.to_device(text)
    parts, params, hists, tots = [], [], [], []
    for r in range(world):                                   # pass 1 per shard
        lo, hi = shard.entry_range(n, r, world)
        b = ctx.qv_batch(d_text, ctx.to_device(c.off[lo:hi]), ctx.to_device(c.len[lo:hi]), hi - lo, text_bytes=len(c.text))
        p = ctx.qv_prescan(b, entry0=lo)
        params.append((p.delChar, p.del_first, p.subChar, p.sub_first))
        parts.append((lo, hi, b))
    dC, dF, sC, sF = shard.merge_params(params)
    gp = L.QVParams(dC, sC, dF, sF)
    for lo, hi, b in parts:
        h, t = ctx.qv_hist(b, gp, entry0=lo)
        hists.append(h); tots.append(t)
    hist, tot = shard.merge_hist(hists, tots)
    coding = api.qv_build(hist, tot, gp)
    ctx.qv_set_coding(coding)
    streams = []
    for lo, hi, b in parts:                                  # pass 2 per shard
        m = hi - lo
        blob, hoff, _ = api.frame_headers(c.hdr[lo:hi], None, shard.previous_well(c.hdr, lo))
        d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
        d_rec, d_seg = ctx.alloc(8 * (m + 1)), ctx.alloc(20 * m)
        total = ctx.qv_sizes(b, d_hoff, d_seg, d_rec)
        d_out = ctx.alloc(total)
        ctx.qv_encode(b, d_hdr, d_hoff, d_rec, d_seg, d_out)
        streams.append(d_out.download(np.uint8, total).tobytes())
    head = b"\xaa\x55" + api.qv_write_coding(coding, c.text[: c.text.index(b"/", 1)])
    assert shard.concat(head, streams) == want


# ---- GPU text front end ------------------------------------------------------------------------

@pytest.mark.parametrize("src", ["qv_tiny", "qv_full", "qv_runs", "synth"])
def test_gpu_index_matches_host_index(ctx, src):
    txt = synth.make_quiva(300, seed=12, mean=2500).text if src == "synth" else O.golden(src + ".quiva")
    d = ctx.to_device(np.frombuffer(txt, np.uint8))
    off, ln, hdr, pl = ctx.index_quiva_device(d, len(txt))
    off2, ln2, hdr2, pl2 = api.index_quiva(txt)
    assert (off == off2).all() and (ln == ln2).all() and (hdr == hdr2).all() and pl == pl2


@pytest.mark.parametrize("bad", [
    b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc",                 # no final newline
    b"m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n",                # header missing
    b"@m 1 0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n",               # no slash
    b"@m/1/0_3\nabc\nabc\nabc\nabc\nabc\n",                      # RQ field required
    b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\n",                         # incomplete entry
    b"@m/1/0_3 RQ=0.8\nabc\nabc\nab\nabc\nabc\n",                # ragged
    b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n\n",             # blank header line
    b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n@m/2/0_1 RQ=0.8\na\nb\n",
])
def test_gpu_index_rejects_like_host(ctx, bad):
    """Every image the host indexer rejects is rejected by the GPU front end too (the drivers then
    re-run the host indexer for the reference's exact first message)."""
    lib = L.load()
    cnt, pl, line, ec = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
    assert lib.dx_index_quiva(bad, len(bad), 0, None, None, None, C.byref(cnt), C.byref(pl), C.byref(line), C.byref(ec)) == -3
    d = ctx.to_device(np.frombuffer(bad, np.uint8))
    with pytest.raises(L.DexGPUError) as e:
        ctx.index_quiva_device(d, len(bad))
    assert e.value.code == -3
    assert (e.value.line, e.value.idx_code) == (line.value, ec.value)


@pytest.mark.parametrize("kind", ["fasta", "arrow"])
def test_pack2_of_a_text_that_arrives_in_pieces(ctx, kind):
    """dx_file_pack2_stream (dexta -i / dexar -i, files of 256 MB and more): chunks of whole records through the device, the bytes of
    dx_file_pack2 of the whole text whatever the chunk -- records longer than a chunk, empty reads at a piece's end, a read source
    that hands over a few bytes at a time -- and a malformed text's line number counted from the file's first line."""
    rng = np.random.Generator(np.random.PCG64(3))
    lens = np.concatenate([rng.integers(0, 3000, 150), [0, 0, 40000, 0, 7, 0], rng.integers(1, 900, 60)]).astype(np.uint32)
    f = synth.make_seqfile(kind, len(lens), seed=12, lens=lens, width=70)
    want = O.dexta(f.text) if kind == "fasta" else O.dexar(f.text)

    def run(text, chunk, dribble=0):
        pos, parts = [0], {}

        def read(n):
            k = min(n, dribble) if dribble else n
            out = text[pos[0]: pos[0] + k]
            pos[0] += len(out)
            return out
        total = ctx.pack2_stream(read, lambda data, at: parts.__setitem__(at, data) and None, arrow=kind == "arrow", chunk=chunk)
        img = b"".join(parts[a] for a in sorted(parts))
        assert len(img) == total and sorted(parts)[0] == 0
        return img

    for chunk in (4096, 30000, 1 << 20, 1 << 26):
        assert run(f.text, chunk) == want, chunk
    assert run(f.text, 9000, dribble=777) == want
    one = synth.make_seqfile(kind, 1, seed=2, mean=50000, width=80)           # one record, many chunks' worth
    assert run(one.text, 4096) == (O.dexta(one.text) if kind == "fasta" else O.dexar(one.text))
    # a line that is no header where one must be, far into the text: the same refusal, the line counted from the top
    lines_ = f.text.split(b"\n")
    hdrs = [i for i, ln in enumerate(lines_) if ln.startswith(b">")]
    bad = b"\n".join(lines_[:hdrs[150]] + [lines_[hdrs[150]].replace(b"/", b"_")] + lines_[hdrs[150] + 1:])
    with pytest.raises(L.DexGPUError) as e1:
        ctx.dexta(bad) if kind == "fasta" else ctx.dexar(bad)
    with pytest.raises(L.DexGPUError) as e2:
        run(bad, 8192)
    assert e1.value.code == e2.value.code
    def outcome(fn):
        try:
            return fn()
        except L.DexGPUError as e:
            return e.code
    assert outcome(lambda: run(b"", 4096)) == outcome(lambda: ctx.dexta(b"") if kind == "fasta" else ctx.dexar(b""))      # an empty text: as a whole file

    # ... and the other way: the image in pieces, a record cut anywhere (its head, its bases), the text of the whole image
    mode = L.DX_LETTERS_UPPER if kind == "fasta" else L.DX_LETTERS_ARROW
    text = ctx.undexta(want, upper=True, width=70) if kind == "fasta" else ctx.undexar(want, width=70)

    def back(img, chunk, dribble=0):
        pos, parts = [0], {}

        def read(n):
            k = min(n, dribble) if dribble else n
            out = img[pos[0]: pos[0] + k]
            pos[0] += len(out)
            return out
        total = ctx.unpack2_pieces(read, lambda data, at: parts.__setitem__(at, data) and None, mode=mode, width=70, chunk=chunk)
        got = b"".join(parts[a] for a in sorted(parts))
        assert len(got) == total
        return got
    for chunk in (4096, 5000, 1 << 20):
        assert back(want, chunk) == text, chunk
    assert back(want, 4096, dribble=333) == text
    assert outcome(lambda: back(want[:-3], 4096)) == outcome(lambda: ctx.undexta(want[:-3], upper=True, width=70) if kind == "fasta" else ctx.undexar(want[:-3], width=70))
    assert outcome(lambda: back(b"\x00\x01" + want[2:], 4096)) == -3                    # (no endian key: DX_E_FORMAT)


def test_dexqv_large_file_uses_gpu_index_and_still_matches(ctx, monkeypatch):
    c = synth.make_quiva(120, seed=17, mean=9000)                  # > 1 MiB: GPU-indexed in dx_file_dexqv
    assert len(c.text) > (1 << 20)
    a = ctx.dexqv(c.text)
    set_flag(monkeypatch, "host_index", "1")
    b = ctx.dexqv(c.text)
    assert a == b == O.dexqv(c.text)
    bad = c.text[:-5]                                              # truncated last line: same error either way
    with pytest.raises(L.DexGPUError) as e1:
        ctx.dexqv(bad)
    set_flag(monkeypatch, "host_index", None)
    with pytest.raises(L.DexGPUError) as e2:
        ctx.dexqv(bad)
    assert str(e1.value) == str(e2.value)


def _dev(k):
    """Device of the k-th context of a sharded run: distinct GPUs when the box has several, else device 0."""
    return k % max(1, L.load().dx_device_count())


def test_file_dexqv_sharded_cut_beyond_shard0(ctx):
    """~250 k symbols over 4 contexts: the 100000-symbol threshold of QV.c:1006 lies beyond shard 0, whose
    prefix batch must then supply subChar (csrc/dx_files.c)."""
    c = synth.make_quiva(28, seed=17, mean=9000)
    cs = [api.Context(_dev(k)) for k in range(4)]
    try:
        assert api.dexqv_sharded(cs, c.text, 0) == O.dexqv(c.text, 0)
    finally:
        for x in cs:
            x.close()


@pytest.mark.parametrize("nctx", [2, 5])
@pytest.mark.parametrize("lossy", [0, 1])
def test_file_dexqv_sharded_over_contexts(ctx, nctx, lossy):
    """dx_file_dexqv_sharded: host threads + host-side merge; one context per GPU when several are
    visible, otherwise all on device 0."""
    c = synth.make_quiva(43, seed=71, mean=7000)
    cs = [api.Context(_dev(k)) for k in range(nctx)]
    try:
        got = api.dexqv_sharded(cs, c.text, lossy)
    finally:
        for x in cs:
            x.close()
    assert got == O.dexqv(c.text, lossy)
    tiny = synth.make_quiva(3, seed=72, mean=100)                 # fewer entries than contexts
    cs = [api.Context(_dev(k)) for k in range(nctx)]
    try:
        assert api.dexqv_sharded(cs, tiny.text, lossy) == O.dexqv(tiny.text, lossy)
    finally:
        for x in cs:
            x.close()


@pytest.mark.parametrize("nctx", [2, 5])
def test_file_dexqv_sharded_by_byte_ranges(ctx, monkeypatch, nctx):
    """A file too large to index on one thread first (BASELINE configs[4]): dx_file_dexqv_sharded deals BYTES, every shard counts the
    newlines of its range, finds the records that begin in it (six lines a record, data lines may begin with '@' too), uploads and
    indexes them on its own device (dx_index_quiva_device) -- no pass over the whole file anywhere.  DEXGPU_TEST=shard_bytes_min brings a
    small file that way: the reference's bytes; a malformed file comes back with what the one-context driver says about it; a file
    whose first 100000 symbols reach beyond shard 0 goes the serial way and comes out right."""
    set_flag(monkeypatch, "shard_bytes_min", "4096")
    c = synth.make_quiva(320, seed=73, mean=5000)
    cs = [api.Context(_dev(k)) for k in range(nctx)]
    try:
        for lossy in (0, 1):
            assert api.dexqv_sharded(cs, c.text, lossy) == O.dexqv(c.text, lossy)
        short = synth.make_quiva(9000, seed=74, lens=np.full(9000, 37, np.uint32))        # many entries in every range
        assert api.dexqv_sharded(cs, short.text, 0) == O.dexqv(short.text, 0)
        small = synth.make_quiva(24, seed=75, mean=9000)                                  # (100000 symbols reach beyond shard 0 of 5)
        assert api.dexqv_sharded(cs, small.text, 0) == O.dexqv(small.text, 0)
        bad = bytearray(c.text)
        at = c.text.index(b"\n", len(c.text) // 2)
        del bad[at - 3: at]                                                               # a data line three symbols short
        with pytest.raises(L.DexGPUError) as e1:
            ctx.dexqv(bytes(bad))
        with pytest.raises(L.DexGPUError) as e2:
            api.dexqv_sharded(cs, bytes(bad), 0)
        import re
        where = lambda e: (e.value.code, re.search(r"line (\d+)", str(e.value)).group(1), re.search(r"code (\d+)", str(e.value)).group(1))
        assert where(e1) == where(e2), (str(e1.value), str(e2.value))
        cut = c.text[: c.text.rindex(b"\n", 0, len(c.text) - 1) + 1]                      # the last entry a line short
        with pytest.raises(L.DexGPUError) as e3:
            api.dexqv_sharded(cs, cut, 0)
        assert e3.value.code == -3
    finally:
        for x in cs:
            x.close()


def test_sharded_file_drivers_over_every_physical_device():
    """dx_file_dexqv_sharded / dx_file_pack2_sharded with ONE context on EVERY device hipGetDeviceCount reports: the
    hipSetDevice-per-thread path of csrc/dx_files.c on real multi-GPU hardware (BASELINE configs[4]'s layout: contiguous
    entry ranges per GPU, host-side histogram sum, outputs concatenated), against the oracle.  Skips on a one-GPU box --
    there the same drivers run over several contexts of device 0 (the tests around this one)."""
    ndev = L.load().dx_device_count()
    if ndev < 2:
        pytest.skip("one GPU visible (%d): the multi-device path needs at least two" % ndev)
    cs = [api.Context(k) for k in range(ndev)]
    try:
        assert sorted(x.device for x in cs) == list(range(ndev))
        for lossy in (0, 1):
            c = synth.make_quiva(40 * ndev + 3, seed=91, mean=6000)        # ~250 k symbols per device: the 100000-symbol cut lies in shard 0
            assert api.dexqv_sharded(cs, c.text, lossy) == O.dexqv(c.text, lossy)
        small = synth.make_quiva(2 * ndev, seed=92, mean=4000)              # ... and beyond shard 0 here
        assert api.dexqv_sharded(cs, small.text, 0) == O.dexqv(small.text, 0)
        for kind in ("fasta", "arrow"):
            f = synth.make_seqfile(kind, 50 * ndev + 1, seed=93, mean=5000)
            assert api.pack2_sharded(cs, f.text, arrow=(kind == "arrow")) == (O.dexta(f.text) if kind == "fasta" else O.dexar(f.text))
    finally:
        for x in cs:
            x.close()


@pytest.mark.parametrize("nctx", [2, 5])
@pytest.mark.parametrize("kind", ["fasta", "arrow"])
def test_file_pack2_sharded_over_contexts(ctx, nctx, kind):
    """dx_file_pack2_sharded: read ranges on several contexts (here all on device 0), one host thread
    each; the image is byte-identical to the single-context one and to the reference's."""
    lens = np.array([0, 3, 900, 17, 20000] + [int(x) for x in np.random.default_rng(3).integers(1, 6000, 70)], np.uint32)
    c = synth.make_seqfile(kind, len(lens), seed=41, lens=lens)
    others = [api.Context(_dev(k + 1)) for k in range(nctx - 1)]
    try:
        got = api.pack2_sharded([ctx] + others, c.text, arrow=(kind == "arrow"))
    finally:
        for o in others:
            o.close()
    assert got == (O.dexta(c.text) if kind == "fasta" else O.dexar(c.text))
    with pytest.raises(L.DexGPUError):
        api.pack2_sharded([ctx, ctx], b"no header\nACGT\n")


def test_in_memory_entry_api(ctx):
    """dx_entries_* (QVcoding_Scan1 / Compress_Next_QVentry1 shape): bare records + per-entry offsets
    equal the oracle's per-entry encodes with the tables of the same scan."""
    c = synth.make_quiva(33, seed=88, mean=7000)
    text = np.frombuffer(c.text, np.uint8)
    lib = L.load()
    e = lib.dx_entries_new()
    lines_of = []
    for i in range(len(c.len)):
        Ln, o = int(c.len[i]), int(c.off[i])
        lines = [text[o + k * (Ln + 1): o + k * (Ln + 1) + Ln].tobytes() for k in range(5)]
        lines_of.append(lines)
        assert lib.dx_entries_add(e, Ln, *lines) == 0
    coding, rec, nb, coff = L.QVCoding(), C.c_void_p(), C.c_size_t(), C.c_void_p()
    rc = lib.dx_entries_compress(ctx.h, e, 0, C.byref(coding), C.byref(rec), C.byref(nb), C.byref(coff))
    assert rc == 0
    got = C.string_at(rec.value, nb.value)
    offs = np.ctypeslib.as_array(C.cast(coff, C.POINTER(C.c_uint64)), (len(c.len) + 1,)).copy()
    lib.dx_file_free(rec); lib.dx_file_free(coff); lib.dx_entries_free(e)
    ref = O.qv_create(O.qv_scan(c.text))
    assert (coding.delChar, coding.subChar) == (ref.delChar, ref.subChar)
    at = 0
    for i, lines in enumerate(lines_of):
        body, _ = O.qv_encode_entry(ref, False, np.stack([np.frombuffer(x, np.uint8) for x in lines]))
        assert int(offs[i]) == at and got[at: at + len(body)] == body
        at += len(body)
    assert at == len(got) == int(offs[-1])


@pytest.mark.parametrize("kind", ["fasta", "arrow"])
def test_gpu_seq_index_matches_host_index(ctx, kind):
    for txt in (O.golden("ta_edge.fasta") if kind == "fasta" else O.golden("ar_edge.arrow"),
                synth.make_seqfile(kind, 200, seed=13, mean=3000, width=70).text):
        d = ctx.to_device(np.frombuffer(txt, np.uint8))
        got = ctx.index_seq_device(d, len(txt), arrow=(kind == "arrow"))
        want = api.index_seq(txt, arrow=(kind == "arrow"))
        for a, b in zip(got[:4], want[:4]):
            assert (np.asarray(a) == np.asarray(b)).all()
        if kind == "arrow":
            assert (got[4] == want[4]).all()
        assert got[5] == want[5]


@pytest.mark.parametrize("kind,bad", [
    ("fasta", b"m/1/0_3 RQ=0.8\nACG\n"), ("fasta", b">m 1 0_3 RQ=0.8\nACG\n"), ("fasta", b">m/1/0_3 RQ=0.8\nACG"),
    ("fasta", b">m/1/0_3 RQ=0.8\nACG\n>m/x\nAC\n"), ("arrow", b">m/1/0_3 SN=1.0,2.0\n123\n"), ("fasta", b""),
])
def test_gpu_seq_index_rejects_like_host(ctx, kind, bad):
    with pytest.raises(L.DexGPUError):
        api.index_seq(bad, arrow=(kind == "arrow"))
    d = ctx.to_device(np.frombuffer(bad + b" ", np.uint8))
    with pytest.raises(L.DexGPUError) as e:
        ctx.index_seq_device(d, len(bad), arrow=(kind == "arrow"))
    assert e.value.code == -3


@pytest.mark.parametrize("kind", ["fasta", "arrow"])
def test_pack2_large_file_uses_gpu_index(ctx, kind, monkeypatch):
    c = synth.make_seqfile(kind, 150, seed=19, mean=9000)
    assert len(c.text) > (1 << 20)
    f = ctx.dexta if kind == "fasta" else ctx.dexar
    a = f(c.text)
    set_flag(monkeypatch, "host_index", "1")
    assert a == f(c.text) == (O.dexta(c.text) if kind == "fasta" else O.dexar(c.text))


def _fixed_coding(sym_len, run_len, run_esc, del_char, sub_char):
    """A hand-made prefix code: every symbol sym_len bits, every run run_len bits; run 255 is the
    escape of a type-2 run scheme when run_esc."""
    cd = L.QVCoding()
    for k in range(6):
        sc = cd.s[k]
        ln = sym_len if k < 4 else run_len
        sc.type = 2 if (k >= 4 and run_esc) else 0
        for x in range(256):
            sc.lens[x] = ln
            sc.bits[x] = (x * 73 + k) & 0xff if ln == 8 else ((x * 257 + 3 * k) & ((1 << ln) - 1))
    cd.delChar, cd.subChar = del_char, sub_char
    return cd


@pytest.mark.parametrize("sym_len,run_len,run_esc", [(16, 16, True), (8, 13, False), (8, 14, False), (9, 9, True),
                                                     (3, 9, False)])
@pytest.mark.parametrize("run_p", [0.05, 0.7])
def test_qv_encode_with_given_coding(ctx, sym_len, run_len, run_esc, run_p):
    """dx_qv_set_coding with a caller's tables (the Read_QVcoding situation): fixed-length codes put
    a lane's string exactly at, below and above the 128 bits of the packing fast path, in plain and
    run-coded streams; long runs take the 16-bit literal."""
    prof = synth.pacbio_profile(del_run_p=run_p, sub_run_p=run_p)
    lens = np.array([1, 15, 16, 17, 63, 64, 1023, 1024, 1025, 2500, 6000, 6001, 9000, 400, 3, 0, 7777], np.uint32)
    c = synth.make_quiva(len(lens), seed=5, lens=lens, prof=prof)
    st = O.qv_scan(c.text)
    del_char = st.delChar
    sub_char = st.subChar if st.subChar >= 0 else int(np.argmax(O.hist_array(st)[3]))   # too few symbols for a scan to keep it
    assert del_char >= 0
    txt, o = bytearray(c.text), int(c.off[12])                    # runs of 600: the 16-bit literal
    txt[o + 100: o + 700] = bytes([del_char]) * 600
    txt[o + 9001 + 100: o + 9001 + 700] = b"N" * 600
    txt[o + 4 * 9001 + 2000: o + 4 * 9001 + 2600] = bytes([sub_char]) * 600
    c.text = bytes(txt)
    cd = _fixed_coding(sym_len, run_len, run_esc, del_char, sub_char)
    ref_cd = O.Coding()
    C.memmove(C.byref(ref_cd), C.byref(cd), C.sizeof(cd))
    b, keep = _upload_quiva(ctx, c)
    ctx.qv_set_coding(cd)
    n = len(lens)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
    total = ctx.qv_sizes(b, None, d_seg, d_rec)
    rec = d_rec.download(np.uint64)
    d_out = ctx.alloc(max(total, 4))
    ctx.qv_encode(b, None, None, d_rec, d_seg, d_out)
    out = d_out.download(np.uint8, total).tobytes()
    seg = d_seg.download(np.uint32, 5 * n).reshape(n, 5)
    text = np.frombuffer(c.text, np.uint8)
    for i in range(n):
        Ln, o = int(c.len[i]), int(c.off[i])
        lines = np.stack([text[o + k * (Ln + 1): o + k * (Ln + 1) + Ln] for k in range(5)])
        body, want_seg = O.qv_encode_entry(ref_cd, False, lines)
        assert list(seg[i]) == want_seg
        assert out[int(rec[i]): int(rec[i + 1])] == body


def test_dexqv_dense_token_stretches(ctx):
    """Skewed (Fibonacci-weighted) alphabets with type-2 run schemes, and 900-symbol stretches without
    a single run character: steps whose run streams need several full passes of 6 tokens per lane,
    next to ordinary steps."""
    rng = np.random.Generator(np.random.PCG64(11))
    f = [1, 1]
    while len(f) < 21:
        f.append(f[-1] + f[-2])
    pool = np.concatenate([np.full(cn, 40 + i, np.uint8) for i, cn in enumerate(f)])
    rare = np.array([40, 41, 42, 43], np.uint8)
    prof = synth.pacbio_profile()
    ents = []
    for e in range(60):
        L = int(rng.integers(2500, 6000))
        b = synth.qv_lines(5, e, L, prof)
        run = rng.random(L) < 0.55
        b[0] = np.where(run, ord("2"), rng.choice(pool, L))
        b[1] = np.where(run, ord("N"), rng.choice(np.frombuffer(b"ACGT", np.uint8), L))
        b[2] = rng.choice(pool, L)
        b[3] = rng.choice(pool, L)
        b[4] = np.where(rng.random(L) < 0.6, ord("z"), rng.choice(pool, L))
        if e % 3 == 1:                                   # dense rare stretches, no run characters inside
            s0 = int(rng.integers(0, L - 900))
            for r in (0, 2, 3, 4):
                b[r, s0:s0 + 900] = rng.choice(rare, 900)
            b[1, s0:s0 + 900] = rng.choice(np.frombuffer(b"acgt", np.uint8), 900)
        ents.append((10 + e * 7, [b[r].tobytes() for r in range(5)]))
    txt = _quiva(ents)
    want = O.dexqv(txt)
    assert ctx.dexqv(txt) == want
    assert ctx.undexqv(want, upper=False) == O.undexqv(want, upper=False)


def _two_pass(ctx, c, coding, lossy=False, given=None):
    n = len(c.len)
    b, keep = given if given is not None else _upload_quiva(ctx, c)
    ctx.qv_set_coding(coding, lossy)
    blob, hoff, _ = api.frame_headers(c.hdr)
    d_hdr, d_hoff = ctx.to_device(blob), ctx.to_device(hoff)
    d_rec, d_seg = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n)
    total = ctx.qv_sizes(b, d_hoff, d_seg, d_rec)
    d_out = ctx.alloc(max(total, 4))
    ctx.qv_encode(b, d_hdr, d_hoff, d_rec, d_seg, d_out)
    return (total, d_out.download(np.uint8, total).tobytes(), d_rec.download(np.uint64).copy(),
            d_seg.download(np.uint32, 5 * n).copy(), (b, keep, d_hdr, d_hoff))


@pytest.mark.parametrize("tokens", ["tokens", "tokens_direct", "text"])
@pytest.mark.parametrize("groups", [None, "3", "8"])
@pytest.mark.parametrize("case", ["pacbio", "small_lengths", "dense", "sparse", "lossy", "long_codes", "no_runs", "huge_entry",
                                  "odd_entries"])
def test_encode_onepass_equals_two_pass(ctx, case, groups, tokens, monkeypatch):
    """dx_qv_encode_onepass gives the bytes of the oracle, entry by entry, and the bytes, record offsets and
    segment index of dx_qv_sizes + dx_qv_encode.  Three routes, all writing the records in place: from the tokens
    k_qv_hist left for the batch, the sizes from the entries' own histograms (k_qv_encode_fast; the generic kernel
    for the entries whose tokens are unusable) -- the product path; from the tokens, the sizes from tokens and plain
    lines (k_qv_sizes_fast: DEXGPU_TEST=sizes_from_tokens); and without tokens, from the text alone."""
    if groups:                                                    # several groups: two streams
        set_flag(monkeypatch, "onepass_groups", groups)
    if tokens == "text":
        set_flag(monkeypatch, "no_tokens", "1")
    if tokens == "tokens_direct":
        set_flag(monkeypatch, "sizes_from_tokens", "1")
    lossy = case == "lossy"
    if case == "small_lengths":
        lens = np.array(list(range(0, 70)) + [1023, 1024, 1025, 2047, 4097, 0, 1, 9000], np.uint32)
        c = synth.make_quiva(len(lens), seed=5, lens=lens)
    elif case == "huge_entry":
        lens = np.array([300, 140000, 5, 70001, 0, 2000], np.uint32)
        c = synth.make_quiva(len(lens), seed=8, lens=lens)
    elif case in ("dense", "sparse"):
        p_ = 0.03 if case == "dense" else 0.995
        c = synth.make_quiva(40, seed=6, mean=5000, prof=synth.pacbio_profile(del_run_p=p_, sub_run_p=p_))
    elif case == "odd_entries":                                   # among ordinary entries: runs of 126 / 127 / 300 (the token's run
        c = synth.make_quiva(40, seed=9, mean=4000)               # field ends at 126), a byte >= 128, a line without run characters
        txt, rc_d = bytearray(c.text), O.qv_scan(c.text).delChar
        def put(e, line, at, data):
            o_ = int(c.off[e]) + line * (int(c.len[e]) + 1) + at
            txt[o_: o_ + len(data)] = data
        for e, run in ((3, 126), (5, 127), (9, 300)):
            put(e, 0, 49, b"5" + bytes([rc_d]) * run + b"5"); put(e, 1, 49, b"A" + b"N" * run + b"A")
        put(12, 0, 7, bytes([200])); put(12, 1, 7, b"C")
        put(14, 4, 100, bytes([131, 132]))
        ln17 = int(c.len[17])
        put(17, 0, 0, bytes(34 + (k % 11) for k in range(ln17))); put(17, 1, 0, b"ACGT" * (ln17 // 4) + b"A" * (ln17 % 4))
        c.text = bytes(txt)
    else:
        c = synth.make_quiva(90, seed=7, mean=6000)
    st = O.qv_scan(c.text)
    b, keep = _upload_quiva(ctx, c)
    if case == "long_codes":
        sub = st.subChar if st.subChar >= 0 else int(np.argmax(O.hist_array(st)[3]))
        coding = _fixed_coding(16, 16, True, st.delChar, sub)
    elif case == "no_runs":
        coding = _fixed_coding(7, 9, False, -1, -1)
    else:
        p = ctx.qv_prescan(b)
        hist, tot = ctx.qv_hist(b, p)
        coding = api.qv_build(hist, tot, p, lossy)
    total, out, rec, seg, (b, keep, d_hdr, d_hoff) = _two_pass(ctx, c, coding, lossy, given=(b, keep))
    # tokens for exactly this batch under the run characters of the coding in force (also for the hand-made codings)
    ctx.qv_hist(b, L.QVParams(coding.delChar, coding.subChar, 0, 0))
    n = len(c.len)
    d_rec2, d_seg2, d_out2 = ctx.alloc(8 * (n + 1)), ctx.alloc(20 * n), ctx.alloc(total + 64)
    total2 = ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg2, d_rec2, d_out2, total + 64)
    assert total2 == total
    rec2, seg2 = d_rec2.download(np.uint64), d_seg2.download(np.uint32, 5 * n).reshape(n, 5)
    out2 = d_out2.download(np.uint8, total).tobytes()
    # the one-pass encoder (the product path of the file drivers) directly against the oracle, entry by entry
    ref_cd = O.Coding()
    C.memmove(C.byref(ref_cd), C.byref(coding), C.sizeof(coding))
    blob, hoff, _ = api.frame_headers(c.hdr)
    text = np.frombuffer(c.text, np.uint8)
    at = 0
    for i in range(n):
        Ln, o = int(c.len[i]), int(c.off[i])
        lines = np.stack([text[o + k * (Ln + 1): o + k * (Ln + 1) + Ln] for k in range(5)])
        body, want_seg = O.qv_encode_entry(ref_cd, lossy, lines)
        want_rec = blob[int(hoff[i]): int(hoff[i + 1])].tobytes() + body
        assert int(rec2[i]) == at and list(seg2[i]) == want_seg
        assert out2[at: at + len(want_rec)] == want_rec
        at += len(want_rec)
    assert at == total2 == int(rec2[n])
    # ... and against the two-pass encoder
    assert (rec2 == rec).all()
    assert (seg2.reshape(-1) == seg).all()
    assert out2 == out
    with pytest.raises(L.DexGPUError) as e:                      # too small an output buffer is reported, not overrun
        ctx.qv_encode_onepass(b, d_hdr, d_hoff, d_seg2, d_rec2, d_out2, max(total - 1, 0))
    assert e.value.code == -8


def test_encode_begin_end_pipelines_two_batches(ctx):
    """dx_qv_encode_onepass_begin / _end: batch A's encode is queued, batch B's prescan + hist + table build run while
    A's last compaction may still be going, then A is collected and B encoded the same way: both streams equal the
    one-call encoder's; a second begin before the end, and an end without a begin, are refused."""
    ca, cb = synth.make_quiva(300, seed=41, mean=6000), synth.make_quiva(260, seed=42, mean=7000)
    def prep(c):
        b, keep = _upload_quiva(ctx, c)
        blob, hoff, _ = api.frame_headers(c.hdr)
        n = len(c.len)
        return dict(c=c, b=b, keep=keep, n=n, d_hdr=ctx.to_device(blob), d_hoff=ctx.to_device(hoff),
                    d_rec=ctx.alloc(8 * (n + 1)), d_seg=ctx.alloc(20 * n), d_out=ctx.alloc(len(c.text)), cap=len(c.text))
    A, B = prep(ca), prep(cb)
    def tables(X):
        p = ctx.qv_prescan(X["b"])
        hist, tot = ctx.qv_hist(X["b"], p)
        ctx.qv_set_coding(api.qv_build(hist, tot, p))
    def args(X):
        return (X["b"], X["d_hdr"], X["d_hoff"], X["d_seg"], X["d_rec"], X["d_out"], X["cap"])
    want = {}
    for name, X in (("A", A), ("B", B)):                            # the one-call encoder
        tables(X)
        t = ctx.qv_encode_onepass(*args(X))
        want[name] = (t, X["d_out"].download(np.uint8, t).tobytes(), X["d_rec"].download(np.uint64).copy())
        X["d_out"].upload(np.zeros(X["cap"], np.uint8))
    with pytest.raises(L.DexGPUError):
        ctx.qv_encode_onepass_end()                                # nothing has begun
    tables(A)
    ctx.qv_encode_onepass_begin(*args(A))
    with pytest.raises(L.DexGPUError):
        ctx.qv_encode_onepass_begin(*args(A))                      # one encode at a time
    tables(B)                                                      # the next batch's scan beside A's last compaction
    ta = ctx.qv_encode_onepass_end()
    ctx.qv_encode_onepass_begin(*args(B))
    tb = ctx.qv_encode_onepass_end()
    for name, X, t in (("A", A, ta), ("B", B, tb)):
        assert t == want[name][0]
        assert X["d_out"].download(np.uint8, t).tobytes() == want[name][1]
        assert (X["d_rec"].download(np.uint64) == want[name][2]).all()
    assert ctx.dexqv(ca.text) == O.dexqv(ca.text)                   # (the file driver still works on this context)


def test_record_offsets_beyond_4gib(ctx):
    """A few thousand huge entries (700 k symbols each, 17.5 GB of text generated on the device): the record
    stream grows past 4 GiB, so every 64-bit offset path is exercised (slot offsets, record offsets, the
    compaction, the decoder's addresses).  Checked through size-independent properties: offsets strictly
    increasing with the segment index adding up, the on-device decode of the first and last entries equal to
    the text, and the LAST two records (offsets > 4 GiB) decoded by the oracle with the batch's own tables."""
    torch = pytest.importorskip("torch")
    try:                                                            # (torch only lends this test its device memory)
        free = torch.cuda.mem_get_info()[0]
    except RuntimeError as e:
        pytest.skip("torch cannot open the GPU in this process: %s" % e)
    if free < 60 * 2**30:
        pytest.skip("needs ~60 GB of free device memory")
    n, Ln, seed, movie = 5000, 700_000, 11, "m000_000"
    hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1
    lens = np.full(n, Ln, np.uint32)
    hdr4 = synth.headers(n, seed, lens, 0)
    rec_bytes = hlen + 5 * (lens.astype(np.uint64) + 1)
    off = (np.concatenate([[0], np.cumsum(rec_bytes)[:-1]]) + hlen).astype(np.uint64)
    text_bytes = int(rec_bytes.sum())
    prof = synth.pacbio_profile()

    class Ptr:
        def __init__(self, t): self.t, self.ptr = t, t.data_ptr()
    d_text = torch.empty(text_bytes + 64, dtype=torch.uint8, device="cuda")
    t_off, t_len = torch.from_numpy(off.view(np.int64)).cuda(), torch.from_numpy(lens.view(np.int32)).cuda()
    p_text, p_off, p_len = Ptr(d_text), Ptr(t_off), Ptr(t_len)
    ctx.synth_quiva(seed, 0, n, p_off, p_len, Ptr(torch.from_numpy(hdr4.reshape(-1)).cuda()),
                    Ptr(torch.from_numpy(prof.table().reshape(-1)).cuda()), prof.del_run, movie, p_text)
    ctx.sync()
    b = ctx.qv_batch(p_text, p_off, p_len, n, text_bytes=text_bytes + 64)
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    assert tot == n * Ln
    coding = api.qv_build(hist, tot, p)
    ctx.qv_set_coding(coding)
    blob, hoff, _ = api.frame_headers(hdr4)
    p_hdr, p_hoff = Ptr(torch.from_numpy(blob.copy()).cuda()), Ptr(torch.from_numpy(hoff.view(np.int64)).cuda())
    p_rec = Ptr(torch.empty(n + 1, dtype=torch.int64, device="cuda"))
    p_seg = Ptr(torch.empty(5 * n, dtype=torch.int32, device="cuda"))
    cap = int(hoff[-1]) + api.qv_out_bound(hist, n, coding)
    p_out = Ptr(torch.empty(cap + 64, dtype=torch.uint8, device="cuda"))
    total = ctx.qv_encode_onepass(b, p_hdr, p_hoff, p_seg, p_rec, p_out, cap)
    rec = p_rec.t.cpu().numpy().astype(np.uint64)
    seg = p_seg.t.cpu().numpy().astype(np.uint64).reshape(n, 5)
    assert total == int(rec[-1]) > 2**32 and total <= cap
    assert (np.diff(rec) == seg.sum(axis=1) + np.diff(hoff)).all()         # offsets = running sum of framing + segments
    # on-device decode of the first and the last 8 entries
    for a0 in (0, n - 8):
        lo, hi = int(off[a0]) - hlen, int(off[a0 + 7]) + 5 * (Ln + 1)
        d_back = torch.zeros(hi - lo + 64, dtype=torch.uint8, device="cuda")
        o_rel = Ptr(torch.from_numpy((off[a0:a0 + 8] - np.uint64(lo)).view(np.int64)).cuda())
        ctx.qv_decode(p_out, Ptr(p_rec.t[a0:]), Ptr(p_hoff.t[a0:]), Ptr(p_seg.t[5 * a0:]), Ptr(t_len[a0:]), 8, True, Ptr(d_back), o_rel)
        ctx.sync()
        assert int((d_back[: hi - lo] != d_text[lo:hi]).sum()) == 8 * hlen   # only the (unwritten) header bytes differ
    # the last two records through the oracle
    body = p_out.t[int(rec[n - 2]): int(rec[n])].cpu().numpy().tobytes()
    img = b"\xaa\x55" + api.qv_write_coding(coding, ("@" + movie).encode()) + body
    txt = O.undexqv(img, upper=True)
    want = d_text[int(off[n - 2]) - hlen: int(off[n - 1]) + 5 * (Ln + 1)].cpu().numpy().tobytes()
    strip = lambda b_: [ln for ln in b_.split(b"\n") if not ln.startswith(b"@")]
    assert strip(txt) == strip(want)


def test_old_name_shims_like_dex2db(ctx, tmp_path):
    """include/dexcompat.h: QVcoding_Scan1 / Create_QVcoding / Write_QVcoding / Compress_Next_QVentry1 /
    Free_QVcoding called the way dex2DB.c:511-643 calls them (reset, one scan call per read, create, write the
    coding with the caller's prefix, one compress call per read in the same order): the file written equals
    Write_QVcoding's image + the oracle's per-entry encodes with the tables of the same scan."""
    lib = L.load()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.malloc.restype = C.c_void_p

    class QVcoding(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("delScheme", "insScheme", "mrgScheme", "subScheme", "dRunScheme", "sRunScheme")] + \
                   [("delChar", C.c_int), ("subChar", C.c_int), ("flip", C.c_int), ("prefix", C.c_void_p)]
    lib.Create_QVcoding.restype = C.POINTER(QVcoding)
    lib.Write_QVcoding.argtypes = [C.c_void_p, C.POINTER(QVcoding)]
    lib.QVcoding_Scan1.argtypes = [C.c_int] + [C.c_char_p] * 5
    lib.Compress_Next_QVentry1.argtypes = [C.c_int] + [C.c_char_p] * 5 + [C.c_void_p, C.POINTER(QVcoding), C.c_int]
    lib.Free_QVcoding.argtypes = [C.POINTER(QVcoding)]

    c = synth.make_quiva(29, seed=31, mean=8000)
    text = np.frombuffer(c.text, np.uint8)
    ents = []
    for i in range(len(c.len)):
        Ln, o = int(c.len[i]), int(c.off[i])
        ents.append([text[o + k * (Ln + 1): o + k * (Ln + 1) + Ln].tobytes() for k in range(5)])
    lib.QVcoding_Scan1(0, None, None, None, None, None)            # dex2DB.c:511
    for lines in ents:
        lib.QVcoding_Scan1(len(lines[0]), *lines)                  # dex2DB.c:554
    coding = lib.Create_QVcoding(0)                                # dex2DB.c:557
    prefix = b".qvs"
    buf = libc.malloc(len(prefix) + 1)                             # the caller mallocs the prefix (dex2DB.c:561-565)
    C.memmove(buf, prefix + b"\0", len(prefix) + 1)
    coding.contents.prefix = buf
    path = str(tmp_path / "x.qvs").encode()
    f = libc.fopen(path, b"wb")
    lib.Write_QVcoding(f, coding)                                  # dex2DB.c:566
    for lines in ents:
        lib.Compress_Next_QVentry1(len(lines[0]), *lines, f, coding, 0)   # dex2DB.c:619
    libc.fclose(f)
    dC, sC = coding.contents.delChar, coding.contents.subChar
    lib.Free_QVcoding(coding)                                      # frees the prefix too
    got = open(path, "rb").read()

    ref = O.qv_create(O.qv_scan(c.text))
    assert (dC, sC) == (ref.delChar, ref.subChar)
    cd = L.QVCoding()
    C.memmove(C.byref(cd), C.byref(ref), C.sizeof(cd))
    want = api.qv_write_coding(cd, prefix)
    for lines in ents:
        body, _ = O.qv_encode_entry(ref, False, np.stack([np.frombuffer(x, np.uint8) for x in lines]))
        want += body
    assert got == want


def test_file_pointer_shims_in_the_order_of_dexqv(tmp_path):
    """include/dexcompat.h: QVcoding_Scan(FILE *) / Create_QVcoding / Write_QVcoding / Read_Lines + QVentry / Compress_Next_QVentry(FILE *)
    / Free_QVcoding, called by a driver written for this test (tests/compat_driver/dexqv_like.c) in the order of dexqv.c:81-141:
    the file it writes is the REAL dexqv's, byte for byte, for every .quiva golden (lossy ones too, L = 0, no delChar, ...)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "dexqv_like")
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(root, "include"), "-o", exe,
                           os.path.join(root, "tests", "compat_driver", "dexqv_like.c"),
                           "-L" + libdir, "-ldexgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    ran = 0
    for case in O.cases("quiva"):
        if case.get("expect_error"):
            continue
        src = tmp_path / (case["input"] + ".quiva")
        src.write_bytes(O.golden(case["input"] + ".quiva"))
        out = tmp_path / (case["name"] + ".dexqv")
        r = subprocess.run([exe] + (["-l"] if "-l" in case["flags"] else []) + [str(src), str(out)], capture_output=True, timeout=120)
        assert r.returncode == 0, (case["name"], r.stderr[-500:])
        assert out.read_bytes() == O.golden(case["name"] + ".dexqv"), case["name"]
        ran += 1
    assert ran >= 5


def test_the_references_own_mains_over_the_shims(tmp_path):
    """oracle/_ref_compat/{dexqv,undexqv}: the reference's OWN dexqv.c / undexqv.c (main() and all, with its DB.c), compiled
    unchanged against include/dexcompat.h in place of QV.h and linked with libdexgpu.so in place of QV.c (oracle/Makefile:
    compat).  Its main() drives QVcoding_Scan / Create_QVcoding / Write_QVcoding / Compress_Next_QVentry (dexqv.c:81-141) and
    Read_QVcoding / Uncompress_Next_QVentry (undexqv.c:112-208) -- the shims, the GPU behind them -- and must leave the golden
    bytes: every .quiva golden -> its .dexqv, and back (-U) -> the reference's own round trip."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = {t: os.path.join(root, "oracle", "_ref_compat", t) for t in ("dexqv", "undexqv")}
    if not all(os.path.isfile(e) for e in exe.values()):
        pytest.skip("oracle/_ref_compat not built (make -C oracle compat needs /root/reference)")
    ran = 0
    for case in O.cases("quiva"):
        if case.get("expect_error"):
            continue
        d = tmp_path / case["name"]
        d.mkdir()
        src = d / "x.quiva"
        src.write_bytes(O.golden(case["input"] + ".quiva"))
        r = subprocess.run([exe["dexqv"], "-k"] + case["flags"] + [str(src)], capture_output=True, timeout=180, cwd=str(d))
        assert r.returncode == 0, (case["name"], r.stderr[-500:])
        got = (d / "x.dexqv").read_bytes()
        assert got == O.golden(case["name"] + ".dexqv"), case["name"]
        src.unlink()
        r = subprocess.run([exe["undexqv"], "-k", "-U", str(d / "x.dexqv")], capture_output=True, timeout=180, cwd=str(d))
        assert r.returncode == 0, (case["name"], r.stderr[-500:])
        rt = O.golden(case["input"] + ".quiva") if case["rt_is_input"] else O.golden(case["name"] + ".rt.quiva")
        assert src.read_bytes() == rt, case["name"]
        ran += 1
    assert ran >= 5


@pytest.mark.parametrize("name,kind", [("ta_small.legacy", "dexta"), ("ta_small.swapped", "dexta"),
                                       ("ta_small.legacy_swapped", "dexta"), ("ar_small.swapped", "dexar")])
def test_unpack2_older_and_other_endian_layouts_golden(ctx, name, kind):
    """.dexta with key 0x33cc (uint16 fields), byte-swapped 0xcc33 / 0xaa55, byte-swapped .dexar (undexta.c:140-159,
    211-240; undexar.c:138-145) against what the REAL undexta / undexar print for these images (tests/golden)."""
    img = O.golden(f"{name}.{kind}")
    want = O.golden(name + (".rt.fasta" if kind == "dexta" else ".rt.arrow"))
    if kind == "dexta":
        assert ctx.undexta(img, True, 80) == want
        assert ctx.undexta(img, False, 61) == O.undexta(img, False, 61)
    else:
        assert ctx.undexar(img, 80) == want
        assert ctx.undexar(img, 7) == O.undexar(img, 7)


@pytest.mark.parametrize("arrow", [False, True], ids=["bps", "arw"])
def test_pack2_bare_payloads_like_dex2db(ctx, arrow):
    """dx_pack2_encode with d_hdr == NULL: bare 2-bit payloads, read after read, no framing -- what dex2DB writes to
    .bps / .arw through Compress_Read (dex2DB.c:604-606, 643-644) -- against Number_Read / Number_Arrow +
    Compress_Read per read.  In-memory reads: no line ends, every length mod 4, empty reads, odd letters."""
    rng = np.random.Generator(np.random.PCG64(77 + arrow))
    lens = [0, 1, 2, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 4099, 20000, 0, 7] + [int(x) for x in rng.integers(1, 9000, 60)]
    alpha = np.frombuffer(b"1234G05" if arrow else b"ACGTacgtNn", np.uint8)
    reads = [alpha[rng.integers(0, len(alpha), L)].tobytes() for L in lens]
    text = b"".join(reads)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    nsym = np.array(lens, np.uint32)
    want = [O.compress_read(r, arrow) for r in reads]
    out_off = np.concatenate([[0], np.cumsum([len(w) for w in want])]).astype(np.uint64)
    d_text = ctx.to_device(np.frombuffer(text + b"\0" * 16, np.uint8))
    d_off, d_n, d_oo = ctx.to_device(off), ctx.to_device(nsym), ctx.to_device(out_off)
    total = int(out_off[-1])
    d_out = ctx.to_device(np.full(total + 64, 0xEE, np.uint8))
    ctx.pack2_encode(L.DX_ALPHA_ARROW if arrow else L.DX_ALPHA_BASES, d_text, d_off, d_n, d_n, len(lens), None, None, d_out, d_oo)
    got = d_out.download(np.uint8, total + 64).tobytes()
    assert got[:total] == b"".join(want)
    assert got[total:] == b"\xee" * 64                                 # nothing written past the last payload


def _with_runs(c, entry, line, at, run_lens, run_char, gap=b"\x28"):
    """Plants runs of run_char of the given lengths (separated by `gap` symbols) into one QV line of one entry; for the
    deletion line the tag line gets 'N' under the run character and a letter under everything else (dextract.c:99-101)."""
    txt = bytearray(c.text)
    L, o = int(c.len[entry]), int(c.off[entry])
    p = o + line * (L + 1) + at
    for r in run_lens:
        assert p + r + len(gap) <= o + line * (L + 1) + L
        txt[p: p + r] = bytes([run_char]) * r
        txt[p + r: p + r + len(gap)] = gap
        if line == 0:
            q = p + (L + 1)
            txt[q: q + r] = b"N" * r
            txt[q + r: q + r + len(gap)] = b"ACGT"[: len(gap)] if len(gap) <= 4 else b"A" * len(gap)
        p += r + len(gap)
    c.text = bytes(txt)


def test_dexqv_long_runs_stay_on_the_token_path(ctx):
    """Runs of 127 and more do not fit a token's 7-bit run field: they go by the line's exception list and the entry
    stays with k_qv_encode_fast (round 2 sent such entries to the text-reading encoder).  One exception per lane
    (patched in place), two and more within a lane's 8 tokens (token-by-token pass), runs across step boundaries,
    255 / 256 (the code's cap, QV.c:479-482), > 65535 (16-bit literal overflow, QV.c:411, 420), at the line's end."""
    lens = np.array([9000] * 10 + [200000, 9000, 9000], dtype=np.uint32)
    c = synth.make_quiva(len(lens), seed=91, lens=lens)
    st = O.qv_scan(c.text)
    dch, sch = st.delChar, st.subChar
    assert dch >= 0
    sch = sch if sch >= 0 else ord("?")
    _with_runs(c, 1, 0, 100, [127, 126, 128, 300, 5, 255, 256, 1000], dch)                # one exception here and there
    _with_runs(c, 2, 0, 50, [130] * 20, dch)                                               # consecutive exception tokens: several per lane
    _with_runs(c, 2, 4, 900, [127] * 12 + [2000, 127, 127], sch)
    _with_runs(c, 3, 4, 0, [1023, 1, 1024, 2, 1025], sch)                                  # around the 1 KiB step
    _with_runs(c, 4, 0, 9000 - 600, [599], dch, gap=b"")                                   # the line ENDS in a long run
    _with_runs(c, 10, 0, 1000, [60000, 127, 65535], dch)                                   # up to the 16-bit literal's limit
    for beyond in (False, True):
        if beyond:                                                     # beyond 16 bits the reference writes a stream its own
            _with_runs(c, 10, 4, 5, [65536, 127, 70000], sch)          # decoder misreads (QV.c:411, 420): same bytes, no decode
        want = O.dexqv(c.text)
        got = ctx.dexqv(c.text)
        assert len(got) == len(want) and got == want
        assert ctx.qv_onepass_info()["text_entries"] == 0             # nobody went to the text-reading encoder
        if not beyond:
            assert ctx.undexqv(got, upper=True) == c.text
        if O.have_ref():
            import tempfile
            with tempfile.TemporaryDirectory() as d:
                assert O.run_ref("dexqv", [], c.text, ".quiva", ".dexqv", d) == got


@pytest.mark.parametrize("run_p", [0.95, 0.99, 0.999])
def test_dexqv_high_run_density_takes_no_text_entries(ctx, run_p):
    prof = synth.pacbio_profile(del_run_p=run_p, sub_run_p=run_p)
    c = synth.make_quiva(60, seed=78, mean=12000, prof=prof)
    assert ctx.dexqv(c.text) == O.dexqv(c.text)
    assert ctx.qv_onepass_info()["text_entries"] == 0
    # the group index of such lines (decoder side) round-trips too
    back = ctx.undexqv(ctx.dexqv(c.text), upper=True)
    assert back == c.text


def test_undexqv_of_a_bare_file_decodes_with_the_walks_group_index(ctx, monkeypatch):
    """dx_qv_walk_indexed: the host walk of a bare .dexqv leaves the group index the encoder would have left
    (csrc/dx_layout.h), and dx_file_undexqv hands it to the decoder (dx_qv_use_index): the wave-per-line kernels decode a
    file from disk.  Same text as the oracle's, with and without the index; the kernels that ran say which route it was."""
    cases = [synth.make_quiva(n, seed=s, mean=m).text for n, s, m in ((3, 1, 300), (60, 2, 9000), (700, 3, 900), (24, 4, 30000))]
    cases.append(synth.make_quiva(30, seed=5, mean=9000, prof=synth.pacbio_profile(del_run_p=0.999, sub_run_p=0.97)).text)   # long runs
    cases += [O.golden(c["input"] + ".quiva") for c in O.cases("quiva") if "-l" not in c["flags"]]
    for text in cases:
        dx = O.dexqv(text)
        want = O.undexqv(dx, upper=True)
        set_flag(monkeypatch, "walk_index", "1")          # (off by default in the file drivers: it costs the CLI more than it saves)
        ctx.profile(True)
        got = ctx.undexqv(dx, upper=True)
        used = ctx.kernel_times()
        ctx.profile(False)
        assert got == want
        w = api.qv_walk(dx, index=True)
        if w["delChar"] >= 0 or w["subChar"] >= 0:
            assert "k_qv_decode_runs" in used, used.keys()
        assert "k_qv_decode_sub" in used or "k_qv_decode" in used
        set_flag(monkeypatch, "walk_index", None)
        ctx.profile(True)
        assert ctx.undexqv(dx, upper=True) == want
        assert "k_qv_decode_sub" not in ctx.kernel_times() and "k_qv_decode_runs" not in ctx.kernel_times()
        ctx.profile(False)


def test_decode_with_a_host_made_index_equals_the_encoders(ctx):
    """The same stream decoded three ways -- with the encoder's own group index, with the host walk's (dx_qv_use_index),
    without any -- gives the same text; the host walk's index has the size the device layout reserves."""
    c = synth.make_quiva(300, seed=21, mean=7000)
    dx = O.dexqv(c.text)
    w = api.qv_walk(dx, index=True)
    sw = lambda L: (((L + 15) >> 4) + 3) >> 2
    assert int(w["gidx_off"][-1]) == len(w["gidx"]) and w["gidx_none"] == 0
    d_in = ctx.to_device(np.frombuffer(dx, np.uint8).copy())
    d_rec, d_hoff = ctx.to_device(w["rec_off"]), ctx.to_device(w["hdr_off"])
    d_seg, d_len = ctx.to_device(w["seg"].reshape(-1).copy()), ctx.to_device(w["len"])
    lens = w["len"].astype(np.uint64)
    ooff = np.concatenate([[0], np.cumsum(5 * (lens + 1))]).astype(np.uint64)
    d_ooff, d_out = ctx.to_device(ooff[:-1].copy()), ctx.alloc(int(ooff[-1]) + 64)
    coding, _, _, _ = api.qv_read_coding(dx[2:])
    ctx.qv_set_coding(coding, False)
    ctx.qv_decode(d_in, d_rec, d_hoff, d_seg, d_len, w["n"], True, d_out, d_ooff)
    plain = d_out.download(np.uint8, int(ooff[-1]))
    d_gidx, d_goff = ctx.to_device(w["gidx"]), ctx.to_device(w["gidx_off"])
    ctx.qv_use_index(d_in, d_seg, w["n"], d_gidx, d_goff, w["gidx_none"])
    ctx.profile(True)
    ctx.qv_decode(d_in, d_rec, d_hoff, d_seg, d_len, w["n"], True, d_out, d_ooff)
    used = ctx.kernel_times()
    ctx.profile(False)
    ctx.qv_use_index(None, None, 0, None, None)
    assert "k_qv_decode_sub" in used and "k_qv_decode_runs" in used
    assert (d_out.download(np.uint8, int(ooff[-1])) == plain).all()
    body = b"".join(c.text[int(o):int(o) + 5 * (int(l) + 1)] for o, l in zip(c.off, c.len))
    assert bytes(plain) == body


def test_old_name_shim_refuses_other_bytes_in_the_second_pass(tmp_path):
    """Compress_Next_QVentry1 (dexcompat.h) writes the record Create_QVcoding made from what QVcoding_Scan1 was given; the
    reference encodes what it is handed in the second pass (QV.c:1343-1379), so other bytes of the same length must not
    pass silently: the shim checks a checksum of the five lines per entry and dies like on an order mismatch."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, ctypes as C
sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})
import numpy as np
from dextractor_amd import _lib as L, synth
lib = L.load(); libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
lib.Create_QVcoding.restype = C.c_void_p
lib.QVcoding_Scan1.argtypes = [C.c_int] + [C.c_char_p] * 5
lib.Compress_Next_QVentry1.argtypes = [C.c_int] + [C.c_char_p] * 5 + [C.c_void_p, C.c_void_p, C.c_int]
c = synth.make_quiva(6, seed=3, mean=3000)
t = np.frombuffer(c.text, np.uint8)
ents = [[t[int(o) + k * (int(n) + 1): int(o) + k * (int(n) + 1) + int(n)].tobytes() for k in range(5)] for o, n in zip(c.off, c.len)]
lib.QVcoding_Scan1(0, None, None, None, None, None)
for e in ents: lib.QVcoding_Scan1(len(e[0]), *e)
coding = lib.Create_QVcoding(0)
f = libc.fopen({str(tmp_path / 'x.qvs')!r}.encode(), b"wb")
lib.Compress_Next_QVentry1(len(ents[0][0]), *ents[0], f, coding, 0)
bad = list(ents[1]); bad[2] = bytes([bad[2][0] ^ 1]) + bad[2][1:]
print("SECOND", flush=True)
lib.Compress_Next_QVentry1(len(bad[0]), *bad, f, coding, 0)
print("NOT REACHED", flush=True)
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert b"SECOND" in r.stdout and b"NOT REACHED" not in r.stdout
    assert b"not the ones QVcoding_Scan1 was given" in r.stderr


@pytest.mark.parametrize("walk", ["host", "device"])
def test_old_name_decode_shims_like_undexqv(ctx, tmp_path, monkeypatch, walk):
    """include/dexcompat.h: Read_QVcoding / Uncompress_Next_QVentry driven the way undexqv.c:101-208 drives them -- the
    caller reads the 0x55aa key and, per entry, the framing bytes itself through the same FILE*, the shim hands out the
    five lines (lower-case tags) and leaves the stream at the next record.  The text put together that way is the
    reference's `undexqv` output.  (walk: where the shim's plan walks the records -- the host, or the device whatever the
    file's size: its index then comes down from there, dx_file_undexqv_plan_index.)"""
    import struct
    if walk == "device":
        set_flag(monkeypatch, "device_walk_min", "0")
        set_flag(monkeypatch, "walk_piece", "4096")
    lib = L.load()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.fread.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    libc.fread.restype = C.c_size_t

    class QVcoding(C.Structure):
        _fields_ = [(k, C.c_void_p) for k in ("delScheme", "insScheme", "mrgScheme", "subScheme", "dRunScheme", "sRunScheme")] + \
                   [("delChar", C.c_int), ("subChar", C.c_int), ("flip", C.c_int), ("prefix", C.c_char_p)]
    lib.Read_QVcoding.restype = C.POINTER(QVcoding)
    lib.Read_QVcoding.argtypes = [C.c_void_p]
    lib.Uncompress_Next_QVentry.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(QVcoding), C.c_int]
    lib.Free_QVcoding.argtypes = [C.POINTER(QVcoding)]

    def rd(f, n):
        b = C.create_string_buffer(n)
        got = libc.fread(b, 1, n, f)
        return b.raw[:got]

    for text in (synth.make_quiva(17, seed=41, mean=4000).text, O.golden("qv_runs.quiva"), O.golden("qv_full.quiva")):
        dx = O.dexqv(text)
        want = O.undexqv(dx, upper=False)
        path = tmp_path / "x.dexqv"
        path.write_bytes(dx)
        f = libc.fopen(str(path).encode(), b"rb")
        assert rd(f, 2) == b"\xaa\x55"                                  # undexqv.c:103-110
        coding = lib.Read_QVcoding(f)                                   # undexqv.c:112
        prefix = coding.contents.prefix
        out, well = [], 0
        while True:                                                     # undexqv.c:119-208
            b = rd(f, 1)
            if not b:
                break
            while b[0] == 255:
                well += 255
                b = rd(f, 1)
            well += b[0]
            beg, end, qv = struct.unpack("<iii", rd(f, 12))
            rlen = end - beg
            bufs = [C.create_string_buffer(rlen + 1) for _ in range(5)]
            entry = (C.c_char_p * 5)(*[C.cast(x, C.c_char_p) for x in bufs])
            assert lib.Uncompress_Next_QVentry(f, entry, coding, rlen) == 0
            out.append(b"%s/%d/%d_%d RQ=0.%d\n" % (prefix, well, beg, end, qv))
            out += [x.raw[:rlen] + b"\n" for x in bufs]
        lib.Free_QVcoding(coding)
        libc.fclose(f)
        assert b"".join(out) == want


def test_per_read_helpers_of_db_h(ctx):
    """DB.h:257-267 under their own names (include/dexcompat.h), one in-memory string a call, over the 2-bit kernels with
    the alphabets DX_ALPHA_NUMBERS / DX_LETTERS_NUMBERS: Number_Read / Compress_Read / Uncompress_Read / Lower_Read /
    Upper_Read as DB.c:319-416 defines them (every byte value as a letter: the maps of Number_Read, bytes >= 128 -> 0 as in
    test_pack2_arbitrary_bytes; line ends excepted: the packer drops them), Number_Arrow / Letter_Arrow :383, 418-441."""
    lib = L.load()
    rng = np.random.default_rng(77)
    base = np.zeros(256, np.uint8)
    for ch, v in ((b"cC", 1), (b"gG", 2), (b"tT", 3)):
        for c in ch: base[c] = v
    arrow = np.full(256, 3, np.uint8); arrow[ord("1")] = 0; arrow[ord("2")] = 1; arrow[ord("3")] = 2; arrow[ord("G")] = 2
    for n in [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 63, 64, 65, 1000, 1023, 1024, 1025, 9999, 70001]:
        letters = rng.integers(1, 256, n, dtype=np.uint8)
        letters[letters == 10] = ord("a")                               # (no line ends inside a read's string)
        want = base[letters]
        buf = C.create_string_buffer(letters.tobytes() + b"\0", n + 8)
        lib.Number_Read(buf)
        got = np.frombuffer(buf.raw[: n + 1], np.uint8)
        assert (got[:n] == want).all() and got[n] == 4
        lib.Compress_Read(n, buf)
        clen = (n + 3) >> 2
        pad = np.concatenate([want, np.zeros(4 * clen - n, np.uint8)]).reshape(-1, 4)
        packed = (pad[:, 0] << 6) | (pad[:, 1] << 4) | (pad[:, 2] << 2) | pad[:, 3]
        assert (np.frombuffer(buf.raw[:clen], np.uint8) == packed).all()
        lib.Uncompress_Read(n, buf)
        got = np.frombuffer(buf.raw[: n + 1], np.uint8)
        assert (got[:n] == want).all() and got[n] == 4
        lib.Upper_Read(buf)
        assert buf.raw[: n + 1] == bytes(b"ACGT"[v] for v in want) + b"\0"
        lib.Number_Read(buf); lib.Lower_Read(buf)
        assert buf.raw[: n + 1] == bytes(b"acgt"[v] for v in want) + b"\0"
        al = letters.copy(); al[al == 0] = ord("4")
        buf = C.create_string_buffer(al.tobytes() + b"\0", n + 8)
        lib.Number_Arrow(buf)
        got = np.frombuffer(buf.raw[: n + 1], np.uint8)
        assert (got[:n] == arrow[al]).all() and got[n] == 4
        lib.Letter_Arrow(buf)
        assert buf.raw[: n + 1] == bytes(b"1234"[v] for v in arrow[al]) + b"\0"
    empty = C.create_string_buffer(b"\0", 8)
    lib.Number_Read(empty)
    assert empty.raw[0] == 4
    lib.Lower_Read(empty)
    assert empty.raw[0] == 0
