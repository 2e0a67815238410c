"""CPU-side tests of the product's host logic (no GPU): the C-ABI library loads and exports every
declared symbol; Huffman/coding-header builder, record framing and the text front end agree with
the oracle and with the reference's golden bytes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from _flags import set_flag, test_env

import _oracle as O
from dextractor_amd import _lib as L
from dextractor_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "dexgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in dexgpu.h but not exported"
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    compat = open(os.path.join(ROOT, "include", "dexcompat.h")).read()
    compat = re.sub(r"/\*.*?\*/", "", compat, flags=re.S)
    old_names = set(re.findall(r"\b([A-Z][A-Za-z]*_[A-Za-z_]*[a-z]1?|QVentry)\s*\(", compat))
    assert old_names == {"QVcoding_Scan1", "Create_QVcoding", "Write_QVcoding", "Compress_Next_QVentry1", "Free_QVcoding",
                         "Read_QVcoding", "Uncompress_Next_QVentry",
                         "QVcoding_Scan", "Compress_Next_QVentry", "Read_Lines", "QVentry", "Set_QV_Line", "Get_QV_Line",
                         "Compress_Read", "Uncompress_Read", "Lower_Read", "Upper_Read", "Number_Read", "Letter_Arrow", "Number_Arrow"}, old_names
    for name in old_names:
        assert hasattr(lib, name), f"{name} declared in dexcompat.h but not exported"


def _params(st):
    return L.QVParams(st.delChar, st.subChar, st.del_first, st.sub_first)


def _raw_hist(st):
    h = O.hist_array(st)
    h[4:6] -= 1                       # dx_qv_build adds the reference's start value of 1 itself
    return h


@pytest.mark.parametrize("case", O.cases("quiva"), ids=lambda c: c["name"])
def test_build_matches_oracle_and_golden_header(case):
    txt = O.golden(case["input"] + ".quiva")
    lossy = "-l" in case["flags"]
    st = O.qv_scan(txt)
    want = O.qv_create(st, lossy)
    got = api.qv_build(_raw_hist(st), st.totChar, _params(st), lossy)
    assert (got.delChar, got.subChar) == (want.delChar, want.subChar)
    for s in range(6):
        if (s == 4 and got.delChar < 0) or (s == 5 and got.subChar < 0):
            continue
        assert got.s[s].type == want.s[s].type
        assert list(got.s[s].bits) == list(want.s[s].bits)
        assert list(got.s[s].lens) == list(want.s[s].lens)
    # Write_QVcoding bytes == the golden file's header (after dexqv's own 0x55aa key)
    dx = O.golden(case["name"] + ".dexqv")
    prefix = txt[: txt.index(b"/", 1)]
    img = api.qv_write_coding(got, prefix)
    assert dx[:2] == b"\xaa\x55" and dx[2: 2 + len(img)] == img
    back, flip, pre, used = api.qv_read_coding(dx[2:])
    assert flip == 0 and pre == prefix and used == len(img)
    assert (back.delChar, back.subChar) == (got.delChar, got.subChar)
    assert list(back.s[1].bits) == list(got.s[1].bits) and list(back.s[1].lens) == list(got.s[1].lens)


def test_build_random_histograms_match_oracle():
    rng = np.random.Generator(np.random.PCG64(5))
    lib = O.lib()
    for trial in range(300):
        nsym = int(rng.integers(1, 257))
        kind = trial % 4
        h = np.zeros(256, np.uint64)
        idx = rng.choice(256, nsym, replace=False)
        if kind == 0:
            h[idx] = rng.integers(1, 1000, nsym)
        elif kind == 1:
            h[idx] = rng.integers(1, 4, nsym)                       # many ties
        elif kind == 2:
            h[idx] = (1.6 ** np.minimum(rng.permutation(nsym), 80)).astype(np.uint64) + 1   # deep tree -> truncated
        else:
            h[idx] = 1 << rng.integers(0, 40, nsym).astype(np.uint64)
        hist = np.zeros((6, 256), np.uint64)
        hist[:4] = h
        hist[4:] = rng.integers(0, 5, (2, 256))
        p = L.QVParams(-1, -1, -1, -1)
        try:
            got = api.qv_build(hist, 10, p, False)
        except L.DexGPUError as e:
            assert e.code == -5                                    # a >16-bit escape code: unsupported
            continue
        first, want = O.Scheme(), O.Scheme()
        hh = (C.c_uint64 * 256)(*[int(x) for x in h])
        assert lib.ref_huffman(hh, None, C.byref(first)) == 0
        if first.type:
            assert lib.ref_huffman(hh, C.byref(first), C.byref(want)) == 0
        else:
            want = first
        assert got.s[0].type == want.type
        assert list(got.s[0].bits) == list(want.bits) and list(got.s[0].lens) == list(want.lens)


def test_build_degenerate_and_errors():
    hist = np.zeros((6, 256), np.uint64)
    with pytest.raises(L.DexGPUError) as e:
        api.qv_build(hist, 0, L.QVParams(-1, -1, -1, -1), False)
    assert e.value.code == -4                                      # empty histogram


def test_frame_headers_known_bytes():
    hdr = np.array([[5, 0, 5, 851], [5, 10, 10, 800], [260, 3, 10, 0], [770, 0, 2, 85]], np.int32)
    blob, off, last = api.frame_headers(hdr)
    want = bytes.fromhex("05" "00000000" "05000000" "53030000"
                         "00" "0a000000" "0a000000" "20030000"
                         "ff00" "03000000" "0a000000" "00000000"
                         "ffff00" "00000000" "02000000" "55000000")
    assert blob.tobytes() == want and list(off) == [0, 13, 26, 40, 55] and last == 770
    # negative delta wraps the byte like the reference's (uint8) cast, dexta.c:192
    blob, off, _ = api.frame_headers(np.array([[10, 0, 0, 0], [7, 0, 0, 0]], np.int32))
    assert blob[13] == (7 - 10) & 0xff
    # arrow framing + SNR conversion (dexar.c:159-163)
    lib = L.load()
    assert [lib.dx_snr_to_cnr(np.float32(v)) for v in (6.81, 99.99, 100.5, 0.0)] == [680, 9998, 9999, 0]
    cnr = np.array([[680, 9998, 9999, 0]], np.uint16)
    blob, off, _ = api.frame_headers(np.array([[5, 0, 6, 0]], np.int32), cnr)
    assert blob.tobytes() == bytes.fromhex("05" "00000000" "06000000" "a8020e270f270000")


def test_index_quiva_matches_generator():
    c = synth.make_quiva(9, seed=5, mean=700)
    off, ln, hdr, pl = api.index_quiva(c.text)
    assert (off == c.off).all() and (ln == c.len).all() and (hdr == c.hdr).all()
    assert c.text[:pl] == b"@m000_000"


@pytest.mark.parametrize("bad,code", [
    (b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc", 5),            # no final newline after an entry's LAST line: the reference reads lines 2-5
                                                                    # with fgets and compares strlen, newline included (QV.c:786-795): "not the same
                                                                    # length" (found against the real binary by tools/stress_cli.py; this said 1 before)
    (b"@m/1/0_3 RQ=0.8\nabc", 1),                                    # ... after an entry's FIRST line: "Last line does not end with a newline" (QV.c:779)
    (b"@m/1/0_3 RQ=0.8", 1),                                         # ... after a header line: the same
    (b"m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n", 2),           # header missing
    (b"@m 1 0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabc\n", 3),          # no slash
    (b"@m/1/0_3\nabc\nabc\nabc\nabc\nabc\n", 3),                 # RQ field required (QV.c:964)
    (b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\n", 4),                    # incomplete entry
    (b"@m/1/0_3 RQ=0.8\nabc\nabc\nab\nabc\nabc\n", 5),           # ragged
])
def test_index_quiva_rejects(bad, code):
    lib = L.load()
    cnt, pl, line, ec = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_index_quiva(bad, len(bad), 0, None, None, None, C.byref(cnt), C.byref(pl), C.byref(line), C.byref(ec))
    assert rc == -3 and ec.value == code
    with pytest.raises(ValueError):
        O.dexqv(bad)                                               # the oracle rejects it too


def test_index_quiva_takes_a_last_line_one_character_longer_like_the_reference():
    """An unterminated last line with ONE character more than the entry's other lines has the strlen the reference compares
    (QV.c:792): it passes, its last character standing where the newline would."""
    lib = L.load()
    good = b"@m/1/0_3 RQ=0.8\nabc\nabc\nabc\nabc\nabcd"
    off, ln = (C.c_uint64 * 1)(), (C.c_uint32 * 1)()
    hdr = (C.c_int32 * 4)()
    cnt, pl, line, ec = C.c_uint64(), C.c_size_t(), C.c_uint64(), C.c_int()
    rc = lib.dx_index_quiva(good, len(good), 1, off, ln, hdr, C.byref(cnt), C.byref(pl), C.byref(line), C.byref(ec))
    assert rc == 0 and cnt.value == 1 and ln[0] == 3 and off[0] == 16


@pytest.mark.parametrize("kind", ["fasta", "arrow"])
def test_index_seq_matches_generator(kind):
    c = synth.make_seqfile(kind, 7, seed=3, mean=300, width=70)
    off, tl, ns, hdr, cnr, pl = api.index_seq(c.text, arrow=(kind == "arrow"))
    assert (off == c.off).all() and (tl == c.tlen).all() and (ns == c.len).all()
    assert (hdr[:, :3] == c.hdr[:, :3]).all()
    if kind == "fasta":
        assert (hdr[:, 3] == c.hdr[:, 3]).all()
    else:
        want = (np.array(c.snr, dtype=np.float64) * 100 + 0.5).astype(np.int64)
        assert (cnr == want).all()


def test_index_seq_edge_file():
    txt = O.golden("ta_edge.fasta")
    off, tl, ns, hdr, cnr, pl = api.index_seq(txt)
    assert list(ns) == [5, 0, 7, 2, 4, 1, 6, 173, 80]
    assert list(tl) == [6, 0, 8, 3, 5, 2, 7, 176, 81]
    assert list(hdr[:, 0]) == [5, 5, 260, 770, 771, 1025, 1281, 1881, 1881]
    assert list(hdr[:, 3]) == [851, 800, 0, 85, 9, 75, 123456, 77, 5]
    assert txt[:pl] == b">mv"


@pytest.mark.parametrize("case", O.cases("quiva"), ids=lambda c: c["name"])
def test_walk_bare_file_index(case):
    """dx_qv_walk (host boundary walk of a bare .dexqv) against per-entry oracle encodes."""
    txt = O.golden(case["input"] + ".quiva")
    dx = O.golden(case["name"] + ".dexqv")
    lossy = "-l" in case["flags"]
    w = api.qv_walk(dx)
    off, ln, hdr, pl = api.index_quiva(txt)
    assert w["n"] == len(ln) and (w["len"] == ln).all() and (w["hdr4"] == hdr).all()
    assert w["prefix"] == txt[:pl] and w["newv"] == 1 and w["flip"] == 0
    assert int(w["rec_off"][-1]) == len(dx)
    coding = O.qv_create(O.qv_scan(txt), lossy)
    text = np.frombuffer(txt, np.uint8)
    for i in range(w["n"]):
        L, o = int(ln[i]), int(off[i])
        lines = np.stack([text[o + k * (L + 1): o + k * (L + 1) + L] for k in range(5)])
        body, seg = O.qv_encode_entry(coding, lossy, lines)
        assert list(w["seg"][i]) == seg
        hl = int(w["hdr_off"][i + 1] - w["hdr_off"][i])
        assert dx[int(w["rec_off"][i]) + hl: int(w["rec_off"][i + 1])] == body


@pytest.mark.parametrize("case", O.cases("quiva"), ids=lambda c: c["name"])
def test_undexqv_plan_knows_the_text_size_without_a_gpu(case):
    """dx_file_undexqv_plan (host walk + header lines, undexqv.c:182, 206-207): the size of the reference's output."""
    dx = O.golden(case["name"] + ".dexqv")
    rt = O.golden(case["input"] + ".quiva") if case["rt_is_input"] else O.golden(case["name"] + ".rt.quiva")
    assert api.undexqv_plan_size(dx) == len(rt)
    with pytest.raises(L.DexGPUError) as e:
        api.undexqv_plan_size(dx[: len(dx) // 2])
    assert e.value.code == -3


def test_walk_of_an_image_without_records_with_and_without_index():
    """A .dexqv image that holds a coding but no records (cut at rec_off[0]): the walk finds zero entries either way, and
    the group index of dx_qv_walk_indexed is empty -- not a NULL pointer handed to numpy."""
    img = O.golden("qv_nodel.dexqv")
    w = api.qv_walk(img)
    head = img[: int(w["rec_off"][0])]
    for index in (False, True):
        e = api.qv_walk(head, index=index)
        assert e["n"] == 0 and len(e["rec_off"]) == 1 and len(e["len"]) == 0
        if index:
            assert e["gidx"].dtype == np.uint32 and len(e["gidx"]) == 0 and len(e["gidx_off"]) == 1


def test_walk_rejects_garbage():
    with pytest.raises(L.DexGPUError):
        api.qv_walk(b"\xaa\x55\xcc\x33" + bytes(40))
    dx = O.golden("qv_tiny.dexqv")
    with pytest.raises(L.DexGPUError):
        api.qv_walk(dx[:-7])


def test_walk_byteswapped_file():
    """A file written on a host of the other endianness (keys 0xaa55 / 0xcc33): same index."""
    dx = O.golden("qv_full.dexqv")
    w = api.qv_walk(dx)
    fl = O.byteswap_dexqv(dx, w)
    assert O.undexqv(fl, upper=True) == O.undexqv(dx, upper=True)      # the oracle's flip path agrees
    w2 = api.qv_walk(fl)
    assert w2["flip"] == 1 and w2["n"] == w["n"]
    assert (w2["seg"] == w["seg"]).all() and (w2["hdr4"] == w["hdr4"]).all() and (w2["rec_off"] == w["rec_off"]).all()


def test_walk_on_several_threads_equals_front_to_back(monkeypatch):
    """Large images are walked by several host threads from guessed-and-verified record starts (dx_host.c):
    same index as the front-to-back walk; a damaged image is still rejected."""
    c = synth.make_quiva(900, seed=77, mean=9000)
    dx = O.dexqv(c.text)
    assert len(dx) > 8 << 20
    monkeypatch.setenv("DEXGPU_WALK_THREADS", "1")
    w1 = api.qv_walk(dx)
    monkeypatch.setenv("DEXGPU_WALK_THREADS", "7")
    set_flag(monkeypatch, "walk_require_parallel", "1")
    w7 = api.qv_walk(dx)
    assert w7["n"] == w1["n"] == len(c.len)
    for k in ("rec_off", "hdr_off", "seg", "len", "hdr4"):
        assert (w7[k] == w1[k]).all()
    set_flag(monkeypatch, "walk_require_parallel", None)
    bad = bytearray(dx)
    bad[len(bad) // 2: len(bad) // 2 + 64] = bytes(64)        # a hole in the middle: no chain of walks survives it ...
    got = None
    try:
        got = api.qv_walk(bytes(bad))
    except L.DexGPUError as e:
        assert e.code == -3
    if got is not None:                                       # ... unless the damaged codes still parse: then as front to back
        monkeypatch.setenv("DEXGPU_WALK_THREADS", "1")
        ref = api.qv_walk(bytes(bad))
        assert (got["rec_off"] == ref["rec_off"]).all()
    with pytest.raises(L.DexGPUError):
        api.qv_walk(dx[:-9])


def test_walk_older_layout():
    """No 0x55aa key, uint16 beg/end/qv (undexqv.c:104-109, 159-179): same index, 6 fewer framing bytes per record."""
    dx, leg = O.golden("qv_tiny.dexqv"), O.golden("qv_tiny.legacy.dexqv")
    w, w2 = api.qv_walk(dx), api.qv_walk(leg)
    assert (w2["newv"], w2["flip"], w2["n"]) == (0, 0, w["n"])
    assert (w2["seg"] == w["seg"]).all() and (w2["hdr4"] == w["hdr4"]).all() and (w2["len"] == w["len"]).all()
    assert (np.diff(w["hdr_off"]) - np.diff(w2["hdr_off"]) == 6).all()
    assert int(w2["rec_off"][-1]) == len(leg) and w2["prefix"] == w["prefix"]


def test_out_bound_covers_the_encoded_size():
    """dx_qv_out_bound (sizes d_out of the one-pass encoder) from the raw histograms: never below, and for
    ordinary files close to, the bytes the oracle encodes."""
    for name, lossy in (("qv_full", 0), ("qv_full", 1), ("qv_runs", 0), ("qv_type2", 0), ("qv_nodel", 0), ("qv_tiny", 0)):
        txt = O.golden(name + ".quiva")
        st = O.qv_scan(txt)
        coding = O.qv_create(st, lossy)
        hist = O.hist_array(st).astype(np.uint64).copy()
        hist[4:] -= 1                                         # the oracle's run bins start at 1 (QV.c:934-935); dx_qv_hist's at 0
        cd = L.QVCoding()
        C.memmove(C.byref(cd), C.byref(coding), C.sizeof(cd))
        n = len(api.index_quiva(txt)[1])
        w = api.qv_walk(O.dexqv(txt, lossy))
        body = int(w["rec_off"][-1] - w["rec_off"][0]) - int(w["hdr_off"][-1])
        bound = api.qv_out_bound(hist, n, cd, lossy)
        assert body <= bound
        if name == "qv_full":                                 # (the runs before sub_first are priced at the dearest run token)
            assert bound < 1.25 * body


def test_register_budgets_of_the_kernels_that_share_a_cu():
    """The build leaves every kernel's register / scratch / LDS use in dextractor_amd/kernel_resources.txt.  Four waves per
    SIMD (<= 128 VGPRs) for the kernels whose latency hiding was measured at four; no kernel may spill to scratch.  (Rounds 2
    and 3 held the encoder to 112 so that the compaction's waves fit beside it; round 6 took the scratch slots and their
    compaction kernel out: every route writes its records in place.)"""
    path = os.path.join(os.path.dirname(L.LIB_PATH), "kernel_resources.txt")
    if not os.path.isfile(path):
        pytest.skip("no kernel_resources.txt (library built without the Makefile)")
    res = {}
    for ln in open(path):
        f = ln.split()
        res[f[0]] = {k: int(v) for k, v in (x.split("=") for x in f[1:])}
    def of(prefix):
        hit = [v for k, v in res.items() if k.startswith(prefix)]
        assert hit, prefix
        return hit
    assert all(v["scratch"] == 0 for v in res.values()), {k: v for k, v in res.items() if v["scratch"]}
    assert all(v["vgprs"] <= 128 for v in of("_Z16k_qv_encode_fastILb0EE"))    # the product encoder, no group index
    assert len(of("_Z16k_qv_encode_fast")) == 2                                 # with and without the group index, nothing else
    assert all(v["vgprs"] <= 128 for v in of("_Z11k_qv_encode7qv_args"))         # the generic encoder
    assert not [k for k in res if "k_qv_compact" in k or "k_qv_bounds" in k]      # (the slot route is gone)
    assert all(v["waves_per_simd"] >= 4 for v in of("_Z9k_qv_hist") + of("_Z16k_qv_encode_fast") + of("_Z17k_qv_decode_plain"))
    assert all(v["waves_per_simd"] >= 6 for v in of("_Z15k_qv_decode_subILi2EE"))


def test_no_gpu_means_loud_failure(tmp_path):
    """There is no CPU fallback: without a HIP device the library and the tools refuse to work."""
    import subprocess
    lib = L.load()
    if lib.dx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(L.DexGPUError) as e:
        api.Context(0)
    assert e.value.code == -2 and "no HIP device" in str(e.value)
    fa = tmp_path / "x.fasta"
    fa.write_bytes(b">m/1/0_4 RQ=0.8\nACGT\n")
    r = subprocess.run([os.path.join(ROOT, "dextractor_amd", "bin", "dexta"), "-k", str(fa)], capture_output=True)
    assert r.returncode == 1 and b"cannot open a GPU" in r.stderr
    assert not (tmp_path / "x.dexta").exists() or (tmp_path / "x.dexta").stat().st_size == 0
