"""Sharding arithmetic for one file encoded by several GPUs.

Subreads/entries are independent, so a file is cut into contiguous entry ranges, one per GPU.
For dexqv in one-file mode the ranks exchange -- on the HOST -- the 12 KB histograms (summed) and
the 32-byte order-dependent scan state (first rank that found a delChar wins; subChar comes from
the rank holding entry 0); every rank then builds identical tables.  Outputs are concatenated in
rank order after one copy of the file header.  No RCCL: `torch.distributed` with the gloo backend
(or any host reduction) is enough; `bench.py` and `tests/test_shard_gloo.py` use these helpers.
"""
from __future__ import annotations

import numpy as np


def entry_range(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n_total entries for `rank` of `world`."""
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def merge_params(per_rank):
    """per_rank: list (rank order) of (delChar, del_first, subChar, sub_first) with GLOBAL entry
    indices, -1 where not found.  QV.c:993-1015: delChar belongs to the lowest entry with an n/N
    tag; subChar is decided within the file's first 100000 symbols, i.e. by rank 0's slice."""
    found = [p for p in per_rank if p[0] >= 0]
    d = min(found, key=lambda p: p[1]) if found else (-1, -1, -1, -1)
    s = per_rank[0]
    return int(d[0]), int(d[1]), int(s[2]), int(s[3])


def merge_hist(hists, tots):
    """Sum of the per-shard raw histograms (uint64 [6,256]) and symbol counts."""
    h = np.zeros((6, 256), np.uint64)
    for x in hists:
        h += np.asarray(x, dtype=np.uint64).reshape(6, 256)
    return h, int(sum(int(t) for t in tots))


def previous_well(hdr4_all: np.ndarray, lo: int) -> int:
    """The well the delta chain of a slice starting at entry `lo` continues from (dexqv.c:116)."""
    return 0 if lo == 0 else int(hdr4_all[lo - 1, 0])


def concat(file_header: bytes, shard_streams) -> bytes:
    """A .dexqv/.dexta/.dexar image from its header and the ranks' record streams, in rank order."""
    return file_header + b"".join(shard_streams)
