"""Batches of SHORT entries: the lane-per-entry kernels of dx_qv_short.hpp (k_qs_hist, k_qs_entries) against the oracle.

They are taken for batches of >= 4096 entries none of which is longer than 4096 symbols when the length of the longest entry of a wave,
averaged over the entries, is at most 1000 (fixed lengths: the mean; lognormal lengths: about 1.45 x the mean); every case here checks
that it took them (dx_qv_onepass_info: direct == 3) -- or, where the case is about NOT taking them, that it did not.  Bar: bit-exact."""
import numpy as np
import pytest

from _flags import set_flag, test_env

import _oracle as O
from dextractor_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture
def force(monkeypatch):
    """the lane-per-entry kernels whatever the batch's lengths would cost them (they are for batches whose longest entry's lane is done when
    the batch is: a small batch with a long entry is not theirs) -- the kernels' code for long lines is what these cases are about"""
    set_flag(monkeypatch, "short_force", "1")


def _took_short(ctx):
    return ctx.qv_onepass_info()["direct"] == 3


def _quiva(lines_of, n, movie=b"m7"):
    out = []
    for e in range(n):
        b = lines_of(e)
        L = len(b[0])
        out.append(b"@%s/%d/%d_%d RQ=0.%d\n" % (movie, 10 + 3 * e, 0, L, 800 + e % 100))
        out += [bytes(x) + b"\n" for x in b]
    return b"".join(out)


@pytest.mark.parametrize("lossy", [0, 1])
@pytest.mark.parametrize("seed,n,mean", [(1, 5000, 120), (2, 4500, 500), (3, 20000, 300)])
def test_short_dexqv_vs_oracle(ctx, force, seed, n, mean, lossy):
    c = synth.make_quiva(n, seed=seed, mean=mean)
    if int(c.len.max()) > 4096:                       # (the lognormal tail: clipped so that the batch qualifies)
        c = synth.make_quiva(n, seed=seed, lens=np.minimum(c.len, 4096).astype(np.uint32))
    assert ctx.dexqv(c.text, lossy) == O.dexqv(c.text, lossy)
    assert _took_short(ctx)


def test_short_scan_vs_oracle(ctx):
    """k_qs_hist == QVcoding_Scan's histograms, the run histograms from the entry on in which the run character was found"""
    c = synth.make_quiva(6000, seed=11, dist="fixed", mean=250)
    st = O.qv_scan(c.text)
    d_text = ctx.to_device(np.frombuffer(c.text, np.uint8))
    d_off, d_len = ctx.to_device(c.off), ctx.to_device(c.len)
    b = ctx.qv_batch(d_text, d_off, d_len, len(c.len), text_bytes=len(c.text))
    p = ctx.qv_prescan(b)
    assert (p.delChar, p.subChar, p.del_first, p.sub_first) == (st.delChar, st.subChar, st.del_first, st.sub_first)
    hist, tot = ctx.qv_hist(b, p)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert tot == st.totChar
    for s in range(6):
        assert (hist[s] == want[s]).all(), f"histogram {s}"


def test_short_every_small_length(ctx, force):
    """every length 0..600 and the chunk / request boundaries, several times over"""
    lens = np.array((list(range(0, 601)) + [1007, 1008, 1009, 1023, 1024, 1025, 1040, 2048, 2049, 3000, 4095, 4096]) * 8, dtype=np.uint32)
    c = synth.make_quiva(len(lens), seed=31, lens=lens)
    assert len(lens) >= 4096
    assert ctx.dexqv(c.text) == O.dexqv(c.text)
    assert _took_short(ctx)


@pytest.mark.parametrize("run_p", [0.02, 0.6, 0.97, 0.999])
def test_short_run_densities(ctx, force, run_p):
    prof = synth.pacbio_profile(del_run_p=run_p, sub_run_p=run_p)
    c = synth.make_quiva(4200, seed=77, dist="fixed", mean=700, prof=prof)
    assert ctx.dexqv(c.text) == O.dexqv(c.text)
    assert _took_short(ctx)


def test_short_type2_escapes_and_pad_rule(ctx, force):
    """Fibonacci-weighted symbols (8-bit escapes behind the longest code) in the plain lines, runs beyond 255 (16-bit literals)
    in the run-coded ones: every branch of the pad rule (QV.c:436-442) sees both kinds of last code"""
    rng = np.random.Generator(np.random.PCG64(3))
    f = [1, 1]
    while len(f) < 23:
        f.append(f[-1] + f[-2])
    pool = np.concatenate([np.full(cn, 40 + i, np.uint8) for i, cn in enumerate(f)])
    prof = synth.pacbio_profile()

    def lines_of(e):
        L = int(rng.integers(0, 900))
        b = synth.qv_lines(5, e, L, prof)
        b[2] = rng.choice(pool, L)
        b[3] = rng.choice(pool, L)
        if e % 5 == 0 and L > 400:                     # a run of 300+ of each run character, one of them to the line's end
            b[0, 50:380] = ord("2"); b[1, 50:380] = ord("N")
            b[4, L - 320:] = ord("?")
        return [b[r].tobytes() for r in range(5)]

    txt = _quiva(lines_of, 4300)
    want = O.dexqv(txt)
    assert ctx.dexqv(txt) == want
    assert _took_short(ctx)
    assert ctx.undexqv(want, upper=True) == O.undexqv(want, upper=True)


def test_short_without_run_characters(ctx, force):
    """no N under any deletion QV: no deletion run character (all tags packed, the deletion line coded plain)"""
    rng = np.random.Generator(np.random.PCG64(5))
    prof = synth.pacbio_profile()

    def lines_of(e):
        L = int(rng.integers(1, 500))
        b = synth.qv_lines(8, e, L, prof)
        b[1] = rng.choice(np.frombuffer(b"ACGT", np.uint8), L)
        return [b[r].tobytes() for r in range(5)]

    txt = _quiva(lines_of, 4100)
    assert O.qv_scan(txt).delChar == -1
    assert ctx.dexqv(txt) == O.dexqv(txt)
    assert _took_short(ctx)


def test_short_two_pass_api_gives_the_same_records(ctx, force):
    """dx_qv_sizes + dx_qv_encode (k_qs_entries<false> / <true> on their own) against per-entry oracle calls"""
    c = synth.make_quiva(4100, seed=8, dist="fixed", mean=200)
    st = O.qv_scan(c.text)
    ref_coding = O.qv_create(st)
    d_text = ctx.to_device(np.frombuffer(c.text, np.uint8))
    d_off, d_len = ctx.to_device(c.off), ctx.to_device(c.len)
    b = ctx.qv_batch(d_text, d_off, d_len, len(c.len), text_bytes=len(c.text))
    p = ctx.qv_prescan(b)
    hist, tot = ctx.qv_hist(b, p)
    ctx.qv_set_coding(api.qv_build(hist, tot, p))
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
    for i in range(0, n, 7):
        L = int(c.len[i]); o = int(c.off[i])
        lines = np.stack([text[o + k * (L + 1): o + k * (L + 1) + L] for k in range(5)])
        body, want_seg = O.qv_encode_entry(ref_coding, False, lines)
        hl = int(hoff[i + 1] - hoff[i])
        at = int(rec[i])
        assert list(seg[i]) == want_seg
        assert out[at + hl: at + hl + len(body)] == body
    assert int(rec[n]) == total


def test_long_entry_or_few_entries_keep_the_wave_per_entry_kernels(ctx, monkeypatch):
    c = synth.make_quiva(5000, seed=4, dist="fixed", mean=150)
    want = O.dexqv(c.text)
    set_flag(monkeypatch, "no_short", "1")
    assert ctx.dexqv(c.text) == want and not _took_short(ctx)
    set_flag(monkeypatch, "no_short", None)
    lens = np.full(5000, 150, np.uint32); lens[4321] = 9000
    c = synth.make_quiva(5000, seed=4, lens=lens)
    assert ctx.dexqv(c.text) == O.dexqv(c.text) and not _took_short(ctx)
    c = synth.make_quiva(300, seed=4, dist="fixed", mean=150)
    assert ctx.dexqv(c.text) == O.dexqv(c.text) and not _took_short(ctx)
    c = synth.make_quiva(4500, seed=2, mean=900)           # lognormal: a wave's longest entry is 1.45 x the mean -- more than 1000
    c = synth.make_quiva(4500, seed=2, lens=np.minimum(c.len, 4096).astype(np.uint32))
    assert ctx.dexqv(c.text) == O.dexqv(c.text) and not _took_short(ctx)


def _took_mixed(ctx):
    return ctx.qv_onepass_info()["direct"] == 5


def test_mixed_lengths_1_to_20000(ctx, monkeypatch):
    """a batch of entries of 1 ... 20 000 symbols, most of them short: the lanes take the short ones, the entries of more than 4096 symbols
    go -- a batch of their own -- to the wave-per-entry kernels (direct == 5; round 5 sent the whole batch there for ONE long entry); the
    same batch with the long ones cut down to 4096 (the longest entry's lane would still be running when the rest of so small a batch is
    done: the wave-per-entry kernels) and to 700 (the lane-per-entry kernels alone): all against the oracle"""
    rng = np.random.Generator(np.random.PCG64(20))
    lens = np.concatenate([rng.integers(1, 400, 4000), rng.integers(400, 3000, 700), rng.integers(3000, 20001, 60),
                           [1, 2, 3, 15, 16, 17, 20000, 4096, 4097]]).astype(np.uint32)
    rng.shuffle(lens)
    c = synth.make_quiva(len(lens), seed=21, lens=lens)
    want = O.dexqv(c.text)
    set_flag(monkeypatch, "short_force", "1")                  # (so small a batch: the lanes whatever the longest short entry's lane costs)
    assert ctx.dexqv(c.text) == want and _took_mixed(ctx)
    assert ctx.undexqv(want, upper=True) == O.undexqv(want, upper=True)
    assert ctx.dexqv(c.text, True) == O.dexqv(c.text, True) and _took_mixed(ctx)
    set_flag(monkeypatch, "no_mixed", "1")                     # the switch: one long entry sends the batch to the wave-per-entry kernels
    assert ctx.dexqv(c.text) == want and not _took_mixed(ctx) and not _took_short(ctx)
    set_flag(monkeypatch, "no_mixed", None)
    set_flag(monkeypatch, "short_force", None)
    assert ctx.dexqv(c.text) == want and not _took_short(ctx)   # unforced: a 4096-symbol entry's lane outlasts so small a batch
    c = synth.make_quiva(len(lens), seed=21, lens=np.minimum(lens, 4096).astype(np.uint32))
    assert ctx.dexqv(c.text) == O.dexqv(c.text)
    c = synth.make_quiva(len(lens), seed=21, lens=np.minimum(lens, 700).astype(np.uint32))
    want = O.dexqv(c.text)
    assert ctx.dexqv(c.text) == want and _took_short(ctx)
    assert ctx.undexqv(want, upper=True) == O.undexqv(want, upper=True)


@pytest.mark.parametrize("case", ["usual", "dense_tokens", "no_deletion_run_character", "sizes_from_tokens", "no_tokens", "late_delchar"])
def test_short_entries_with_long_ones_among_them(ctx, monkeypatch, case):
    """The mixed route chosen by itself (20 000 short entries, a few hundred long ones carrying most of the bytes -- what a subread set looks
    like): scan state and histograms (the long entries counted under their own indices: the run histograms start at the entry the run
    character was found in), the oracle's bytes, through every way the long entries' batch can go -- its own histograms, sizes from
    tokens, one run character only, no tokens at all."""
    rng = np.random.Generator(np.random.PCG64(31))
    lens = np.concatenate([rng.integers(100, 600, 20000), np.clip(rng.lognormal(np.log(9000), 0.4, 300), 4097, 60000)]).astype(np.uint32)
    rng.shuffle(lens)
    prof = synth.pacbio_profile(0.4, 0.35) if case == "dense_tokens" else None
    c = synth.make_quiva(len(lens), seed=33, lens=lens, prof=prof)
    if case in ("no_deletion_run_character", "late_delchar"):
        t = bytearray(c.text)
        first = 0 if case == "no_deletion_run_character" else int(np.nonzero(lens > 4096)[0][5]) + 1   # no 'N' tag before that entry: delChar found late
        stop = len(lens) if case == "no_deletion_run_character" else first
        for i in range(stop):
            L_, o = int(c.len[i]), int(c.off[i])
            t[o + L_ + 1: o + 2 * L_ + 1] = bytes(t[o + L_ + 1: o + 2 * L_ + 1]).replace(b"N", b"A").replace(b"n", b"a")
        c.text = bytes(t)
    if case == "sizes_from_tokens":
        set_flag(monkeypatch, "sizes_from_tokens", "1")
    if case == "no_tokens":
        set_flag(monkeypatch, "no_tokens", "1")
    st = O.qv_scan(c.text)
    d_text = ctx.to_device(np.frombuffer(c.text, np.uint8))
    d_off, d_len = ctx.to_device(c.off), ctx.to_device(c.len)
    b = ctx.qv_batch(d_text, d_off, d_len, len(c.len), text_bytes=len(c.text))
    p, hist, tot = ctx.qv_scan(b)
    want = O.hist_array(st)
    want[4:6] -= 1
    assert (p.delChar, p.subChar, p.del_first, p.sub_first) == (st.delChar, st.subChar, st.del_first, st.sub_first)
    assert tot == st.totChar and (hist == want).all()
    for x in (d_text, d_off, d_len): x.free()
    img = ctx.dexqv(c.text)
    assert img == O.dexqv(c.text) and _took_mixed(ctx)
    assert ctx.undexqv(img, upper=True).split(b"\n")[1::6] == c.text.split(b"\n")[1::6]
