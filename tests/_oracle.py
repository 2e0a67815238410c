"""ctypes binding of the ORACLE (oracle/libdexref.so) and fixture helpers -- tests only."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_BIN = os.path.join(ORACLE_DIR, "_ref")

REF_DEL, REF_INS, REF_MRG, REF_SUB, REF_DRUN, REF_SRUN = range(6)


class QVStats(C.Structure):
    _fields_ = [("hist", (C.c_uint64 * 256) * 6), ("totChar", C.c_uint64),
                ("delChar", C.c_int32), ("subChar", C.c_int32),
                ("del_first", C.c_int64), ("sub_first", C.c_int64), ("nentries", C.c_int64)]


class Scheme(C.Structure):
    _fields_ = [("type", C.c_int32), ("bits", C.c_uint32 * 256), ("lens", C.c_int32 * 256)]


class Coding(C.Structure):
    _fields_ = [("s", Scheme * 6), ("delChar", C.c_int32), ("subChar", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "libdexref.so")
        src = os.path.join(ORACLE_DIR, "dexref.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "libdexref.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        u8p = C.c_char_p
        for name in ("ref_dexta", "ref_dexar"):
            getattr(L, name).restype = C.c_long
            getattr(L, name).argtypes = [u8p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_undexta.restype = C.c_long
        L.ref_undexta.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_undexar.restype = C.c_long
        L.ref_undexar.argtypes = [u8p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_dexqv.restype = C.c_long
        L.ref_dexqv.argtypes = [u8p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_undexqv.restype = C.c_long
        L.ref_undexqv.argtypes = [u8p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.ref_qv_scan.restype = C.c_int
        L.ref_qv_scan.argtypes = [u8p, C.c_size_t, C.POINTER(QVStats)]
        L.ref_qv_create.restype = C.c_int
        L.ref_qv_create.argtypes = [C.POINTER(QVStats), C.c_int, C.POINTER(Coding)]
        L.ref_huffman.restype = C.c_int
        L.ref_huffman.argtypes = [C.POINTER(C.c_uint64), C.POINTER(Scheme), C.POINTER(Scheme)]
        L.ref_qv_write_coding.restype = C.c_long
        L.ref_qv_write_coding.argtypes = [C.POINTER(Coding), C.c_char_p, C.c_void_p, C.c_size_t]
        L.ref_qv_encode_entry.restype = C.c_long
        L.ref_qv_encode_entry.argtypes = [C.POINTER(Coding), C.c_int, C.c_int] + [C.c_void_p] * 5 + \
                                         [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        for name in ("ref_number_read", "ref_number_arrow"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [C.c_void_p, C.c_size_t]
        L.ref_compress_read.restype = C.c_size_t
        L.ref_compress_read.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def compress_read(seq: bytes, arrow=False) -> bytes:
    """Number_Read / Number_Arrow + Compress_Read on one in-memory read (DB.c:393-441, 319-338): what dex2DB
    writes to .bps / .arw (dex2DB.c:604-606, 643-644)."""
    num = C.create_string_buffer(bytes(seq), len(seq) + 4)
    (lib().ref_number_arrow if arrow else lib().ref_number_read)(num, len(seq))
    out = C.create_string_buffer((len(seq) + 3) // 4 + 4)
    n = lib().ref_compress_read(len(seq), num, out)
    return out.raw[:n]


def _call(fn, data, cap, *mid):
    buf = C.create_string_buffer(cap)
    n = fn(data, len(data), *mid, buf, cap)
    if n < 0:
        raise ValueError(f"{fn.__name__} -> {n}")
    return buf.raw[:n]


def dexta(txt):   return _call(lib().ref_dexta, txt, len(txt) // 3 + 4096)
def dexar(txt):   return _call(lib().ref_dexar, txt, len(txt) // 3 + 4096)
def undexta(img, upper=False, width=80): return _call(lib().ref_undexta, img, 9 * len(img) + 65536, int(upper), width)
def undexar(img, width=80):              return _call(lib().ref_undexar, img, 9 * len(img) + 65536, width)
def dexqv(txt, lossy=False):             return _call(lib().ref_dexqv, txt, 2 * len(txt) + 65536, int(lossy))
def undexqv(img, upper=False):           return _call(lib().ref_undexqv, img, 12 * len(img) + 65536, int(upper))


def qv_scan(txt):
    st = QVStats()
    r = lib().ref_qv_scan(txt, len(txt), C.byref(st))
    if r < 0:
        raise ValueError(f"ref_qv_scan -> {r}")
    return st


def qv_create(st, lossy=False):
    c = Coding()
    r = lib().ref_qv_create(C.byref(st), int(lossy), C.byref(c))
    if r < 0:
        raise ValueError(f"ref_qv_create -> {r}")
    return c


def qv_encode_entry(coding, lossy, lines):
    """lines: uint8 [5, L] -> (bytes, seg[5])"""
    lines = np.ascontiguousarray(lines, dtype=np.uint8)
    L = lines.shape[1]
    cap = 16 * L + 64
    buf = C.create_string_buffer(cap)
    seg = (C.c_uint32 * 5)()
    ptr = [lines[r].ctypes.data for r in range(5)]
    n = lib().ref_qv_encode_entry(C.byref(coding), int(lossy), L, *ptr, buf, cap, seg)
    if n < 0:
        raise ValueError(f"ref_qv_encode_entry -> {n}")
    return buf.raw[:n], list(seg)


def hist_array(st):
    return np.ctypeslib.as_array(st.hist).reshape(6, 256).copy()


# ---- fixtures ------------------------------------------------------------------------------

def golden(name):
    p = os.path.join(GOLDEN, name)
    if os.path.exists(p):
        with open(p, "rb") as f:
            return f.read()
    with gzip.open(p + ".gz", "rb") as f:
        return f.read()


def cases(kind=None):
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        cs = json.load(f)
    return [c for c in cs if kind is None or c["kind"] == kind]


def hashes():
    with open(os.path.join(GOLDEN, "hashes.json")) as f:
        return json.load(f)


def have_ref():
    return os.path.isfile(os.path.join(REF_BIN, "dexqv"))


def run_ref(tool, flags, src_bytes, src_ext, dst_ext, tmpdir):
    """Run a real reference tool (oracle/_ref) on bytes; returns the produced file's bytes."""
    src = os.path.join(str(tmpdir), "x" + src_ext)
    with open(src, "wb") as f:
        f.write(src_bytes)
    r = subprocess.run([os.path.join(REF_BIN, tool), "-k", *flags, src], capture_output=True)
    if r.returncode != 0:
        raise RuntimeError(f"{tool}: exit {r.returncode}: {r.stderr.decode()}")
    with open(os.path.join(str(tmpdir), "x" + dst_ext), "rb") as f:
        return f.read()


def legacy_dexqv(dx, walk):
    """The same .dexqv in the older layout undexqv still reads (undexqv.c:104-109, 159-179): no 0x55aa
    key in front of the coding's 0x33cc, and beg / end / qv of every record as uint16 instead of int32.
    All three fields of every record must fit 16 bits.  `walk` = api.qv_walk(dx)."""
    import struct
    out = bytearray(dx[2:int(walk["rec_off"][0])])
    for i in range(walk["n"]):
        r0, r1 = int(walk["rec_off"][i]), int(walk["rec_off"][i + 1])
        hl = int(walk["hdr_off"][i + 1] - walk["hdr_off"][i])
        beg, end, qv = struct.unpack("<iii", dx[r0 + hl - 12: r0 + hl])
        assert 0 <= beg < 65536 and 0 <= end < 65536 and 0 <= qv < 65536
        out += dx[r0: r0 + hl - 12] + struct.pack("<HHH", beg, end, qv) + dx[r0 + hl: r1]
    return bytes(out)


def byteswap_dexqv(dx, walk):
    """The same .dexqv as a host of the other endianness would have written it (every uint16 /
    int32 / uint32 field and code word byte-swapped; tag bytes and well bytes unchanged).
    `walk` = api.qv_walk(dx)."""
    b = bytearray(dx)

    def sw(at, n):
        b[at:at + n] = b[at:at + n][::-1]
    at = 0
    sw(at, 2); at += 2                       # 0x55aa
    sw(at, 2); at += 2                       # 0x33cc
    sw(at, 2); at += 2                       # delChar
    sw(at, 2); at += 2                       # subChar
    plen = int.from_bytes(dx[at:at + 4], "little")
    sw(at, 4); at += 4 + plen
    nschemes = 4 + (walk["delChar"] >= 0) + (walk["subChar"] >= 0)
    for _ in range(nschemes):
        at += 1                              # type
        for _ in range(256):
            ln = dx[at]; at += 1
            if ln:
                sw(at, 4); at += 4
    assert at == int(walk["rec_off"][0])
    for i in range(walk["n"]):
        r0 = int(walk["rec_off"][i]); hl = int(walk["hdr_off"][i + 1] - walk["hdr_off"][i])
        for k in range(3):
            sw(r0 + hl - 12 + 4 * k, 4)      # beg, end, qv
        p = r0 + hl
        for k in range(5):
            n = int(walk["seg"][i][k])
            if k != 1:
                for w in range(0, n, 4):
                    sw(p + w, 4)
            p += n
    return bytes(b)


def rewrite_pack2(img, arrow=False, legacy=False, swap=False):
    """A reference-written .dexta / .dexar image (key 0x55aa, int32 fields, this host's byte order) in the other
    layouts the reference's readers accept: `legacy` = key 0x33cc with uint16 beg / end / qv (undexta.c:140-159,
    211-240; .dexta only), `swap` = as a host of the other endianness would have written it (keys read back as
    0xaa55 / 0xcc33; undexar.c:138-145).  Payload and well bytes are byte strings and stay as they are."""
    import struct
    assert img[:2] == b"\xaa\x55" and not (legacy and arrow)
    e = ">" if swap else "<"
    plen = struct.unpack("<i", img[2:6])[0]
    out = bytearray(struct.pack(e + "H", 0x33cc if legacy else 0x55aa) + struct.pack(e + "i", plen) + img[6:6 + plen])
    at = 6 + plen
    while at < len(img):
        w0 = at
        while img[at] == 255:
            at += 1
        at += 1
        out += img[w0:at]
        beg, end = struct.unpack("<ii", img[at:at + 8])
        at += 8
        if arrow:
            cnr = struct.unpack("<4H", img[at:at + 8])
            at += 8
            out += struct.pack(e + "ii4H", beg, end, *cnr)
        else:
            qv = struct.unpack("<i", img[at:at + 4])[0]
            at += 4
            if legacy:
                assert 0 <= beg < 65536 and 0 <= end < 65536 and 0 <= qv < 65536
            out += struct.pack(e + ("HHH" if legacy else "iii"), beg, end, qv)
        clen = (end - beg + 3) >> 2
        out += img[at:at + clen]
        at += clen
    assert at == len(img)
    return bytes(out)
