"""The record walk of a bare .dexqv on the device (dx_qv_walk_device, csrc/dx_qv_walk.hip) against the host walk
(dx_qv_walk, dx_host.c) and the oracle's undexqv: undexqv.c:119-208, QV.c:1428-1481.  The device walk keeps only offsets on an
unbroken chain of lane walks from the first record, so its index must be the host's word for word -- on many pieces and few,
records shorter and longer than a piece, headers with leading 255s, empty entries, every table kind."""
import numpy as np
import pytest

from _flags import set_flag, test_env

import _oracle as O
from dextractor_amd import _lib as L
from dextractor_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with api.Context(0) as c:
        yield c


def _walk_both(ctx, img):
    """-> (host index, device index as numpy, DeviceIndex info)"""
    h = api.qv_walk(img)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    d = ctx.to_device(np.frombuffer(img, np.uint8))
    x = ctx.qv_walk_device(d, len(img), 2 + used, coding, 1, flip)
    try:
        return h, x.download(), (x.pieces, x.piece_bytes)
    finally:
        x.free()
        d.free()


def _same(h, g):
    assert g["n"] == h["n"]
    for k in ("rec_off", "hdr_off", "seg", "len", "hdr4"):
        assert g[k].shape == h[k].shape and (g[k] == h[k]).all(), k


@pytest.mark.parametrize("n,mean,piece", [(1500, 4000, 16384), (400, 300, 4096), (3000, 300, 32768), (60, 9000, 8192), (2500, 2000, 0)])
def test_device_walk_is_the_host_walk(ctx, monkeypatch, n, mean, piece):
    """Many pieces per file (DEXGPU_TEST=walk_piece=<bytes>: tests only; 0 = the product's 32 KiB), records shorter and longer than a piece."""
    if piece:
        set_flag(monkeypatch, "walk_piece", str(piece))
    c = synth.make_quiva(n, seed=100 + n, mean=mean)
    img = ctx.dexqv(c.text)
    assert img == O.dexqv(c.text)
    h, g, (pieces, pb) = _walk_both(ctx, img)
    assert pieces >= 2 and pieces == (len(img) - int(h["rec_off"][0]) + pb - 1) // pb
    _same(h, g)


def test_device_walk_empty_entries_lossy_and_long_runs(ctx, monkeypatch):
    set_flag(monkeypatch, "walk_piece", "4096")
    lens = np.array([0, 1, 2, 15, 16, 17, 0, 0, 1023, 1024, 1025, 5000, 0, 3, 70000, 2, 0], np.uint32)
    cases = [(synth.make_quiva(len(lens), seed=7, lens=lens).text, False),
             (synth.make_quiva(200, seed=8, mean=3000).text, True),
             (synth.make_quiva(30, seed=9, mean=9000, prof=synth.pacbio_profile(del_run_p=0.999, sub_run_p=0.97)).text, False),   # runs of 255 and more: literals
             (synth.make_quiva(40, seed=10, mean=6000, prof=synth.pacbio_profile(del_run_p=0.3, sub_run_p=0.3)).text, False)]
    for text, lossy in cases:
        img = ctx.dexqv(text, lossy)
        assert img == O.dexqv(text, lossy)
        h, g, _ = _walk_both(ctx, img)
        _same(h, g)


def test_device_walk_wells_that_jump(ctx, monkeypatch):
    """Framing bytes of 255 (255 wells each, undexqv.c:124-133) in front of a header -- and bytes of 255 at the end of the record
    before it, which make a start one byte early walk just as well: the chain takes them off again (trim)."""
    set_flag(monkeypatch, "walk_piece", "4096")
    c = synth.make_quiva(600, seed=31, mean=1500)
    hdr = c.hdr.copy()
    hdr[:, 0] = np.cumsum(np.where(np.arange(len(hdr)) % 7 == 3, 700 + 255 * (np.arange(len(hdr)) % 5), 1))     # wells that jump by 255 k + j
    text = b"".join(synth.header_text("quiva", "m000_000", hdr[i], False) + c.text[int(c.off[i]): int(c.off[i]) + 5 * (int(c.len[i]) + 1)]
                    for i in range(len(hdr)))
    img = ctx.dexqv(text)
    assert img == O.dexqv(text)
    h, g, _ = _walk_both(ctx, img)
    _same(h, g)
    assert (np.diff(h["hdr_off"]).astype(np.int64) > 13).sum() > 50        # (headers with leading 255s are in it)


def test_undexqv_plans_on_the_device(ctx, monkeypatch):
    """dx_file_undexqv with the device walk (DEXGPU_TEST=device_walk_min=0: whatever the size; the product asks for 256 MB): the
    oracle's text; the walk kernels ran; DEXGPU_TEST=host_walk takes it back to the host."""
    set_flag(monkeypatch, "device_walk_min", "0")
    set_flag(monkeypatch, "walk_piece", "8192")
    for text, upper in ((synth.make_quiva(300, seed=41, mean=2500).text, True), (synth.make_quiva(50, seed=42, mean=12000).text, False)):
        img = O.dexqv(text)
        want = O.undexqv(img, upper=upper)
        ctx.profile(True)
        assert ctx.undexqv(img, upper=upper) == want
        assert "k_qv_walk" in ctx.kernel_times()
        ctx.profile(False)
    set_flag(monkeypatch, "host_walk", "1")
    ctx.profile(True)
    assert ctx.undexqv(img, upper=False) == O.undexqv(img)
    assert "k_qv_walk" not in ctx.kernel_times()
    ctx.profile(False)


def test_what_the_device_walk_turns_down_goes_to_the_host(ctx, monkeypatch):
    """A stream cut short, or bytes damaged in the middle: DX_E_MISMATCH from the device walk; the file driver then walks on the
    host and reports what the host walk reports.  Records much longer than the guesses' budget (two pieces): the chain is
    walked piece by piece from where it arrives -- same index -- or given up after a few dozen rounds -- same text."""
    set_flag(monkeypatch, "device_walk_min", "0")
    set_flag(monkeypatch, "walk_piece", "4096")
    c = synth.make_quiva(120, seed=51, mean=3000)
    img = O.dexqv(c.text)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    cut = img[: len(img) - 1000]
    d = ctx.to_device(np.frombuffer(cut, np.uint8))
    with pytest.raises(L.DexGPUError) as e:
        ctx.qv_walk_device(d, len(cut), 2 + used, coding, 1, flip)
    assert e.value.code == -7                                                # DX_E_MISMATCH
    d.free()
    with pytest.raises(L.DexGPUError) as e:
        ctx.undexqv(cut)
    assert e.value.code == -3                                                # DX_E_FORMAT, the host walk's
    big = synth.make_quiva(6, seed=52, mean=40000)                           # records of ~55 KB, pieces of 4 KB
    img = O.dexqv(big.text)
    assert ctx.undexqv(img) == O.undexqv(img)
    tiny = synth.make_quiva(4000, seed=53, lens=np.full(4000, 6, np.uint32))  # records of ~40 bytes: more of them in a piece than
    img = O.dexqv(tiny.text)                                                 # a lane's scratch holds (a record per 128 bytes)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    d = ctx.to_device(np.frombuffer(img, np.uint8))
    with pytest.raises(L.DexGPUError) as e:
        ctx.qv_walk_device(d, len(img), 2 + used, coding, 1, flip)
    assert e.value.code == -7
    d.free()
    assert ctx.undexqv(img) == O.undexqv(img)                                # ... and the file driver walks it on the host


def test_device_walk_of_a_byte_swapped_file(ctx, monkeypatch):
    """A file written on a host of the other endianness (GETFLIP, QV.c:553-568): code words and framing fields byte-swapped."""
    import struct
    set_flag(monkeypatch, "walk_piece", "4096")
    c = synth.make_quiva(150, seed=61, mean=2000)
    img = O.dexqv(c.text)
    h = api.qv_walk(img)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    first = 2 + used
    # swap the stream: every framing field and every code word; the tags are bytes.  (The coding in front stays as it is: the
    # device walk is told `flip` by its caller, who has read the coding.)
    out = bytearray(img[:first])
    for i in range(int(h["n"])):
        at = int(h["rec_off"][i]); hb = int(h["hdr_off"][i + 1] - h["hdr_off"][i]); seg = [int(x) for x in h["seg"][i]]
        out += img[at: at + hb - 12]
        out += b"".join(struct.pack(">i", v) for v in struct.unpack("<iii", img[at + hb - 12: at + hb]))
        p = at + hb
        for k, sb in enumerate(seg):
            chunk = img[p: p + sb]
            out += chunk if k == 1 else np.frombuffer(chunk, "<u4").astype(">u4").tobytes()
            p += sb
    sw = bytes(out)
    assert len(sw) == len(img)
    d = ctx.to_device(np.frombuffer(sw, np.uint8))
    x = ctx.qv_walk_device(d, len(sw), first, coding, 1, 1)
    g = x.download()
    x.free(); d.free()
    _same({k: h[k] for k in ("n", "rec_off", "hdr_off", "seg", "len", "hdr4")}, g)


def test_device_walk_of_damaged_streams(ctx, monkeypatch):
    """One byte of a sound image changed, or the image cut: the device walk gives the host walk's index or turns the stream down
    (DX_E_MISMATCH) -- never another index, never a fault --, and the file driver with the device walk forced on returns what it
    returns with the walk on the host: the same text, or the same refusal."""
    set_flag(monkeypatch, "walk_piece", "4096")
    c = synth.make_quiva(150, seed=71, mean=2500)
    img = O.dexqv(c.text)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    first = 2 + used
    rng = np.random.default_rng(5)
    outcomes = {"same": 0, "turned down": 0}
    for k in range(60):
        bad = bytearray(img)
        if k % 6 == 5:
            bad = bad[: int(rng.integers(first + 50, len(img)))]
        else:
            at = int(rng.integers(first, len(img)))
            bad[at] = (bad[at] + int(rng.integers(1, 256))) & 0xff if k % 2 else (255 if bad[at] != 255 else 0)
        bad = bytes(bad)
        d = ctx.to_device(np.frombuffer(bad, np.uint8))
        try:
            x = ctx.qv_walk_device(d, len(bad), first, coding, 1, flip)
        except L.DexGPUError as e:
            assert e.code == -7, e
            outcomes["turned down"] += 1
        else:
            g = x.download()
            x.free()
            h = api.qv_walk(bad)                                            # what the device accepts the host accepts, and finds the same
            _same(h, g)
            outcomes["same"] += 1
        d.free()
        res = []
        for env in ({"device_walk_min": "0"}, {"host_walk": "1"}):
            for kk, vv in env.items(): set_flag(monkeypatch, kk, vv)
            try:
                res.append(ctx.undexqv(bad))
            except L.DexGPUError as e:
                res.append(e.code)
            for kk in env: set_flag(monkeypatch, kk, None)
        assert res[0] == res[1], (k, type(res[0]), type(res[1]))
    assert outcomes["same"] > 5 and outcomes["turned down"] > 5, outcomes


def test_a_walk_that_cannot_allocate_leaves_the_host_walk_a_clean_slate(ctx, monkeypatch):
    """The device walk's scratch does not fit (DEXGPU_TEST=fail_malloc_over=<bytes>: allocations beyond that size fail like a full device):
    dx_qv_walk_device reports it, and the file driver goes on with the host walk -- whose launches must not trip over the failed
    allocation's error (the runtime keeps it until it is read)."""
    set_flag(monkeypatch, "device_walk_min", "0")
    set_flag(monkeypatch, "walk_piece", "4096")
    c = synth.make_quiva(400, seed=81, mean=3000)
    img = O.dexqv(c.text)
    want = O.undexqv(img)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    d = ctx.to_device(np.frombuffer(img, np.uint8))
    set_flag(monkeypatch, "fail_malloc_over", str(len(img) // 8))       # the records' scratch (0.44 of the image) fails, the tables do not
    with pytest.raises(L.DexGPUError) as e:
        ctx.qv_walk_device(d, len(img), 2 + used, coding, 1, flip)
    assert e.value.code in (-2, -6), e.value                                 # DX_E_NOMEM / DX_E_HIP
    set_flag(monkeypatch, "fail_malloc_over", None)
    d.free()
    # through the file driver: the image goes up (it fits), the walk's scratch does not, the host plan takes over
    set_flag(monkeypatch, "fail_malloc_over", str(len(img) // 8))
    set_flag(monkeypatch, "fail_malloc_under", str(len(img) // 2 + len(img) // 4))
    ctx.profile(True)
    got = ctx.undexqv(img)
    assert "k_qv_walk" not in ctx.kernel_times()                             # (the device walk did not get as far as a kernel)
    ctx.profile(False)
    set_flag(monkeypatch, "fail_malloc_over", None)
    set_flag(monkeypatch, "fail_malloc_under", None)
    assert got == want
    set_flag(monkeypatch, "fail_malloc_over", str(len(img) // 8))       # ... and when not even the image goes up
    with pytest.raises(L.DexGPUError):
        ctx.undexqv(img)
    set_flag(monkeypatch, "fail_malloc_over", None)
    assert ctx.undexqv(img) == want                                          # nothing of it is left behind


@pytest.mark.parametrize("n,mean,piece,prof", [(1500, 4000, 16384, None), (60, 9000, 8192, None), (300, 3000, 4096, (0.3, 0.3)),
                                               (40, 9000, 4096, (0.999, 0.97)), (2500, 2000, 0, None), (3000, 40, 4096, None)])
def test_decode_with_the_groups_the_walk_notes(ctx, monkeypatch, n, mean, piece, prof):
    """dx_qv_walk_device notes a word per 8 tokens of every run-coded line (dx_layout.h, DXL_RUN_EIGHTS) and lays them out as
    k_qv_decode_runs takes them (dx_qv_use_dindex): the decode with them is the oracle's text (QV.c:604-691, 823-847), the wave-per-line
    kernel ran, and the lines that got no groups (runs with literals: a group's positions beyond 16 bits) still come out right."""
    if piece:
        set_flag(monkeypatch, "walk_piece", str(piece))
    kw = {"prof": synth.pacbio_profile(del_run_p=prof[0], sub_run_p=prof[1])} if prof else {}
    c = synth.make_quiva(n, seed=300 + n, mean=mean, **kw)
    img = O.dexqv(c.text)
    want = O.undexqv(img)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    h = api.qv_walk(img)
    d = ctx.to_device(np.frombuffer(img + b"\0" * 64, np.uint8))
    x = ctx.qv_walk_device(d, len(img), 2 + used, coding, 1, flip)
    try:
        assert x.n == h["n"] and x.gidx is not None
        L_ = h["len"].astype(np.uint64)
        hl = np.array([len(b"%s/%d/%d_%d RQ=0.%d\n" % (prefix, *h["hdr4"][i])) for i in range(x.n)], np.uint64)
        ooff = (np.cumsum(hl + 5 * (L_ + 1)) - 5 * (L_ + 1)).astype(np.uint64)
        total = int(ooff[-1] + 5 * (L_[-1] + 1)) if x.n else 0
        d_out = ctx.to_device(np.zeros(total + 64, np.uint8))
        d_oo = ctx.to_device(ooff)
        ctx.qv_set_coding(coding)
        x.use(d)
        ctx.profile(True)
        ctx.qv_decode(d, x.rec_off, x.hdr_off, x.seg, x.len, x.n, False, d_out, d_oo)
        kt = ctx.kernel_times()
        ctx.profile(False)
        x.use(None)
        got = d_out.download(np.uint8, total)
        wantb = np.frombuffer(want, np.uint8)
        assert len(wantb) == total
        for i in range(x.n):                                     # (the decoder writes the five data lines; the header lines are the caller's)
            a, b = int(ooff[i]), int(ooff[i] + 5 * (L_[i] + 1))
            assert (got[a:b] == wantb[a:b]).all(), i
        assert "k_qv_decode_runs" in kt
        d_out.free(); d_oo.free()
    finally:
        x.free()
        d.free()


def test_plain_lines_the_walk_leaves_without_words(ctx, monkeypatch):
    """An insertion line of one value nearly throughout codes to a bit a symbol: a burst of the walk's look-ups passes more than
    64 symbols, two marks of the index at once, and the line is left without its words (DXL_SYNC_NONE) -- k_qv_decode_plain takes
    exactly those lines, k_qv_decode_sync the others; same text.  With DEXGPU_TEST=no_syncindex / no_runindex the lane-per-line
    kernels take everything: same text again."""
    import dataclasses
    set_flag(monkeypatch, "walk_piece", "8192")
    base = synth.pacbio_profile()
    flat = dataclasses.replace(base, ins_lut=synth.make_lut([ord("5"), ord("6"), ord("7")], [0.97, 0.02, 0.01]))
    c = synth.make_quiva(300, seed=91, mean=5000, prof=flat)
    img = O.dexqv(c.text)
    want = O.undexqv(img)
    coding, flip, prefix, used = api.qv_read_coding(img[2:])
    h = api.qv_walk(img)
    d = ctx.to_device(np.frombuffer(img + b"\0" * 64, np.uint8))
    x = ctx.qv_walk_device(d, len(img), 2 + used, coding, 1, flip)
    try:
        assert x.n == h["n"] and x.gidx_nosync > 0                          # (lines without their words are in it)
        L_ = h["len"].astype(np.uint64)
        hl = np.array([len(b"%s/%d/%d_%d RQ=0.%d\n" % (prefix, *h["hdr4"][i])) for i in range(x.n)], np.uint64)
        ooff = (np.cumsum(hl + 5 * (L_ + 1)) - 5 * (L_ + 1)).astype(np.uint64)
        total = int(ooff[-1] + 5 * (L_[-1] + 1))
        d_oo = ctx.to_device(ooff)
        ctx.qv_set_coding(coding)
        wantb = np.frombuffer(want, np.uint8)
        for env in ({}, {"no_syncindex": "1"}, {"no_runindex": "1"}):
            for k_, v_ in env.items(): set_flag(monkeypatch, k_, v_)
            d_out = ctx.to_device(np.zeros(total + 64, np.uint8))
            x.use(d)
            ctx.profile(True)
            ctx.qv_decode(d, x.rec_off, x.hdr_off, x.seg, x.len, x.n, False, d_out, d_oo)
            kt = ctx.kernel_times()
            ctx.profile(False)
            x.use(None)
            got = d_out.download(np.uint8, total)
            for i in range(x.n):
                a_, b_ = int(ooff[i]), int(ooff[i] + 5 * (L_[i] + 1))
                assert (got[a_:b_] == wantb[a_:b_]).all(), (env, i)
            if not env:
                assert "k_qv_decode_sub" in kt and "k_qv_decode_plain" in kt and "k_qv_decode_runs" in kt, kt
            d_out.free()
            for k_ in env: set_flag(monkeypatch, k_, None)
        d_oo.free()
    finally:
        x.free()
        d.free()
