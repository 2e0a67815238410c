"""Seeded synthetic .fasta / .arrow / .quiva corpora (BASELINE.json configs).

Every symbol is a pure function of (seed, entry index, stream id, position) through a 32-bit
counter hash, so the numpy generator here and the device generator (csrc/dexgpu_synth.hip)
produce identical bytes without any I/O, and any slice of a corpus can be generated on its own
(per-GPU shards).  Symbols are drawn through 4096-entry lookup tables indexed by the top 12 bits
of the hash; a table therefore *is* the distribution (probabilities in multiples of 1/4096).

The inputs honour the reference's round-trip preconditions: len == end-beg,
wells non-decreasing, RQ without leading zero, one line per QV stream, tag == 'N' exactly where
the deletion QV equals its run character.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

U32 = np.uint32
LUT_BITS = 12
LUT_SIZE = 1 << LUT_BITS

# stream ids (the order of the five lines of a .quiva entry, then fasta / arrow)
S_DEL, S_TAG, S_INS, S_MRG, S_SUB, S_BASE, S_PULSE = range(7)


def lowbias32(x):
    """32-bit integer finalizer (two multiplies, three xor-shifts); x is a uint32 array."""
    x = np.asarray(x, dtype=U32).copy()
    x ^= x >> U32(16)
    x *= U32(0x7FEB352D)
    x ^= x >> U32(15)
    x *= U32(0x846CA68B)
    x ^= x >> U32(16)
    return x


def stream_key(seed: int, entry, stream: int):
    """Per-(entry, stream) 32-bit key; `entry` may be an array of 64-bit indices."""
    e = np.asarray(entry, dtype=np.uint64)
    lo = (e & np.uint64(0xFFFFFFFF)).astype(U32)
    hi = (e >> np.uint64(32)).astype(U32)
    with np.errstate(over="ignore"):
        k0 = lowbias32(U32(seed & 0xFFFFFFFF) + U32(0x9E3779B9) * (lo + U32(1)))
        k1 = lowbias32(k0 ^ (hi * U32(0x85EBCA6B) + U32((stream * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF)))
    return k1


def sample12(key, pos):
    """Top LUT_BITS bits of hash(key, pos); key uint32 (scalar or array), pos uint32 array."""
    with np.errstate(over="ignore"):
        h = lowbias32(np.asarray(key, dtype=U32) + np.asarray(pos, dtype=U32) * U32(0x9E3779B1))
    return (h >> U32(32 - LUT_BITS)).astype(np.intp)


def make_lut(symbols, probs) -> np.ndarray:
    """4096-entry table realising `probs` over `symbols` (largest-remainder rounding)."""
    p = np.asarray(probs, dtype=np.float64)
    p = p / p.sum()
    cnt = np.floor(p * LUT_SIZE).astype(np.int64)
    rem = p * LUT_SIZE - cnt
    for i in np.argsort(-rem, kind="stable")[: LUT_SIZE - cnt.sum()]:
        cnt[i] += 1
    return np.repeat(np.asarray(symbols, dtype=np.uint8), cnt)


def _geom(base, cap, p):
    k = np.arange(cap + 1)
    pr = p * (1 - p) ** k
    pr[-1] = (1 - p) ** cap
    return base + k, pr


@dataclass
class QVProfile:
    """Symbol distributions of the five .quiva streams (SURVEY.md 8(d) config 4 by default)."""
    del_lut: np.ndarray
    ins_lut: np.ndarray
    mrg_lut: np.ndarray
    sub_lut: np.ndarray
    tag_lut: np.ndarray
    del_run: int = ord("2")      # tag is 'N' exactly where del == del_run (dextract.c:99-101); -1: never

    def table(self) -> np.ndarray:
        """[5][4096] uint8 in line order del, tag, ins, mrg, sub (device generator layout)."""
        return np.stack([self.del_lut, self.tag_lut, self.ins_lut, self.mrg_lut, self.sub_lut])


def pacbio_profile(del_run_p=0.85, sub_run_p=0.80) -> QVProfile:
    dsym = np.concatenate([[ord("2")], np.arange(34, 50)])
    dpr = np.concatenate([[del_run_p], np.full(16, (1 - del_run_p) / 16)])
    ssym = np.concatenate([[ord("?")], np.arange(38, 63)])
    spr = np.concatenate([[sub_run_p], np.full(25, (1 - sub_run_p) / 25)])
    return QVProfile(
        del_lut=make_lut(dsym, dpr),
        ins_lut=make_lut(*_geom(33, 20, 0.25)),
        mrg_lut=make_lut(*_geom(33, 60, 0.08)),
        sub_lut=make_lut(ssym, spr),
        tag_lut=make_lut([ord(c) for c in "ACGT"], [1, 1, 1, 1]),
        del_run=ord("2"),
    )


def lengths(n: int, seed: int, dist: str = "lognormal", mean: int = 10000, sigma: float = 0.35,
            lo: int = 500) -> np.ndarray:
    """Per-entry symbol counts (uint32)."""
    if dist == "fixed":
        return np.full(n, mean, dtype=np.uint32)
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    if dist == "short_u":                        # the short part of "mixed" alone: uniform in [100, 600]
        return rng.integers(100, 601, n).astype(np.uint32)
    if dist == "mixed":                          # what a subread set looks like: nine entries in ten short (U[100, 600]), one in ten lognormal around `mean`
        short = rng.integers(100, 601, n).astype(np.int64)
        mu = np.log(mean) - 0.5 * sigma * sigma
        long_ = np.clip(np.rint(rng.lognormal(mu, sigma, n)).astype(np.int64), lo, 20 * mean)
        return np.where(rng.random(n) < 0.1, long_, short).astype(np.uint32)
    if dist == "mixed_tail":                     # (experiments: the same entries, the long ones behind the short ones)
        x = lengths(n, seed, "mixed", mean, sigma, lo)
        return np.concatenate([x[x <= 4096], x[x > 4096]]).astype(np.uint32)
    mu = np.log(mean) - 0.5 * sigma * sigma
    ln = np.rint(rng.lognormal(mu, sigma, n)).astype(np.int64)
    return np.clip(ln, lo, 20 * mean).astype(np.uint32)


def headers(n: int, seed: int, lens: np.ndarray, first_entry: int = 0) -> np.ndarray:
    """int32 [n,4] = well, beg, end, qv for entries [first_entry, first_entry+n).  Wells strictly
    increase by U[1,40) from entry 0 on; end-beg == len; qv in 750..899 (no leading zero after
    'RQ=0.')."""
    tot = first_entry + n
    idx = np.arange(tot, dtype=np.uint64)
    step = 1 + (lowbias32(stream_key(seed, idx, 11)) % U32(39)).astype(np.int64)
    well = np.cumsum(step)[first_entry:]
    idx = idx[first_entry:]
    beg = (lowbias32(stream_key(seed, idx, 12)) % U32(5000)).astype(np.int64)
    qv = 750 + (lowbias32(stream_key(seed, idx, 13)) % U32(150)).astype(np.int64)
    return np.stack([well, beg, beg + np.asarray(lens, dtype=np.int64), qv], axis=1).astype(np.int32)


def qv_lines(seed: int, entry: int, L: int, prof: QVProfile) -> np.ndarray:
    """uint8 [5, L]: del, tag, ins, mrg, sub for one entry."""
    pos = np.arange(L, dtype=U32)
    out = np.empty((5, L), dtype=np.uint8)
    for row, (sid, lut) in enumerate(((S_DEL, prof.del_lut), (S_TAG, prof.tag_lut), (S_INS, prof.ins_lut),
                                      (S_MRG, prof.mrg_lut), (S_SUB, prof.sub_lut))):
        out[row] = lut[sample12(stream_key(seed, entry, sid), pos)]
    if prof.del_run >= 0:
        out[1][out[0] == prof.del_run] = ord("N")
    return out


@dataclass
class Corpus:
    text: bytes                      # the file image
    off: np.ndarray                  # uint64 [n]: offset of the first data line of entry i
    len: np.ndarray                  # uint32 [n]: symbols per entry
    hdr: np.ndarray                  # int32 [n,4]: well, beg, end, qv (arrow: qv unused)
    tlen: np.ndarray = field(default=None)   # uint32 [n]: text bytes of the sequence incl. newlines (fasta/arrow)
    snr: np.ndarray = field(default=None)    # arrow only: the four SN values as printed (strings)


def header_text(kind: str, movie: str, h, fixed_width: bool, snr=None) -> bytes:
    lead = "@" if kind == "quiva" else ">"
    if fixed_width:
        s = f"{lead}{movie}/{h[0]:08d}/{h[1]:07d}_{h[2]:07d}"
    else:
        s = f"{lead}{movie}/{h[0]}/{h[1]}_{h[2]}"
    if kind == "arrow":
        s += " SN=" + ",".join(snr)
    else:
        s += f" RQ=0.{h[3]}"
    return (s + "\n").encode()


def make_quiva(n: int, seed: int = 20261003, dist: str = "lognormal", mean: int = 10000,
               prof: QVProfile | None = None, movie: str = "m000_000", fixed_width: bool = False,
               lens: np.ndarray | None = None, first_entry: int = 0) -> Corpus:
    """.quiva image of entries [first_entry, first_entry+n) of the corpus `seed`."""
    prof = prof or pacbio_profile()
    if lens is None:
        lens = lengths(first_entry + n, seed, dist, mean)[first_entry:]
    hdr = headers(n, seed, lens, first_entry)
    parts, off, at = [], np.empty(n, np.uint64), 0
    nl = np.full((5, 1), 10, dtype=np.uint8)
    for i in range(n):
        h = header_text("quiva", movie, hdr[i], fixed_width)
        L = int(lens[i])
        body = np.concatenate([qv_lines(seed, first_entry + i, L, prof), nl], axis=1).tobytes()
        off[i] = at + len(h)
        parts += [h, body]
        at += len(h) + len(body)
    return Corpus(b"".join(parts), off, lens.astype(np.uint32), hdr)


def safe_snr_values() -> np.ndarray:
    """Two-decimal SN values in [0,99.99] that survive dexar's float32 parse * 100. truncation."""
    v = np.arange(10000)
    txt = np.array([f"{k / 100:.2f}" for k in v])
    f32 = txt.astype(np.float32)
    back = (f32.astype(np.float64) * 100.0).astype(np.int64)
    return txt[back == v]


def make_seqfile(kind: str, n: int, seed: int = 20261003, dist: str = "lognormal", mean: int = 10000,
                 width: int = 80, movie: str = "m000_000", lower: bool = False,
                 lens: np.ndarray | None = None) -> Corpus:
    """.fasta (kind='fasta', uniform ACGT) or .arrow (kind='arrow', pulse widths 1234 with
    p=.45/.30/.15/.10 and round-trip-safe SN values) image, sequence wrapped at `width`."""
    assert kind in ("fasta", "arrow")
    if lens is None:
        lens = lengths(n, seed, dist, mean)
    hdr = headers(n, seed, lens)
    if kind == "fasta":
        lut = make_lut([ord(c) for c in ("acgt" if lower else "ACGT")], [1, 1, 1, 1])
        sid = S_BASE
        safe = None
    else:
        lut = make_lut([ord(c) for c in "1234"], [0.45, 0.30, 0.15, 0.10])
        sid = S_PULSE
        safe = safe_snr_values()
    parts, off, tl, at = [], np.empty(n, np.uint64), np.empty(n, np.uint32), 0
    snr_all = []
    for i in range(n):
        snr = None
        if safe is not None:
            pick = lowbias32(stream_key(seed, np.full(4, i, np.uint64), 20) + np.arange(4, dtype=U32)) % U32(len(safe))
            snr = [str(safe[int(k)]) for k in pick]
            snr_all.append(snr)
        h = header_text(kind, movie, hdr[i], False, snr)
        L = int(lens[i])
        seq = lut[sample12(stream_key(seed, i, sid), np.arange(L, dtype=U32))]
        nlines = (L + width - 1) // width
        body = np.full(L + nlines, 10, dtype=np.uint8)
        idx = np.arange(L)
        body[idx + idx // width] = seq
        off[i] = at + len(h)
        tl[i] = L + nlines
        parts += [h, body.tobytes()]
        at += len(h) + L + nlines
    return Corpus(b"".join(parts), off, lens.astype(np.uint32), hdr, tl,
                  np.array(snr_all) if snr_all else None)
