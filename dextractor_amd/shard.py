"""Sharding arithmetic for one file encoded by several GPUs.

Subreads/entries are independent, so a file is cut into contiguous entry ranges, one per GPU.
For dexqv in one-file mode the ranks exchange -- on the HOST -- the 12 KB histograms (summed) and
the order-dependent scan state (the rank with the lowest n/N tag gives delChar; subChar is the argmax
of the substitution histogram of the entries up to the one where the file's running symbol count
reaches 100000, wherever the rank boundaries fall); every rank then builds identical tables.  Outputs are concatenated in
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


SUB_SCAN_SYMBOLS = 100000      # QV.c:1006: subChar is fixed once the running symbol count reaches this


def merge_del(per_rank):
    """per_rank: list (rank order) of (delChar, del_first) with GLOBAL entry indices, -1 where the
    slice has no n/N tag.  QV.c:993-1002: delChar is the deletion QV under the file's first n/N
    tag, i.e. the answer of the rank with the lowest entry."""
    found = [p for p in per_rank if p[0] >= 0]
    d = min(found, key=lambda p: p[1]) if found else (-1, -1)
    return int(d[0]), int(d[1])


def sub_cut(rank_tots):
    """Which rank holds the entry at which the file's running symbol count first reaches 100000
    (QV.c:1006), given every rank's symbol total in rank order: (rank, symbols still missing when
    that rank's slice starts), or None when the whole file is shorter (no subChar at all)."""
    base = 0
    for r, t in enumerate(rank_tots):
        if base + int(t) >= SUB_SCAN_SYMBOLS:
            return r, SUB_SCAN_SYMBOLS - base
        base += int(t)
    return None


def local_cut(lens, need):
    """Index (within a slice with entry lengths `lens`) of the first entry at which the slice's
    running symbol count reaches `need` (>= 1)."""
    c = np.cumsum(np.asarray(lens, dtype=np.uint64))
    return int(np.searchsorted(c, np.uint64(need), side="left"))


def sub_from_hist(h):
    """QV.c:1010-1013: argmax of the substitution histogram, ties to the smallest value."""
    return int(np.argmax(np.asarray(h)))          # numpy's argmax returns the first maximum


def merge_params(per_rank, rank_tots=None, sub_hists=None, lo_of_rank=None, lens_of_rank=None):
    """Host-side agreement on the order-dependent scan state of QVcoding_Scan (QV.c:993-1015) for
    one file sharded over ranks.  per_rank: (delChar, del_first, subChar, sub_first) per rank,
    GLOBAL entry indices, -1 where not found.

    delChar: lowest entry with an n/N tag wins.  subChar: the argmax of the substitution histogram
    of entries 0..cut, cut = the entry at which the running symbol count of the WHOLE FILE first
    reaches 100000.  If rank 0's slice holds those symbols its own answer is the file's; otherwise
    (small files, many ranks) the caller passes every rank's symbol total (`rank_tots`), the ranks'
    substitution histograms of their entries up to the cut (`sub_hists`: zeros beyond it) and the
    slices' first entries / lengths, and the cut is located here -- what dx_file_dexqv_sharded
    does with its prefix batch (csrc/dx_files.c)."""
    dC, dF = merge_del([(p[0], p[1]) for p in per_rank])
    if rank_tots is None:
        return dC, dF, int(per_rank[0][2]), int(per_rank[0][3])
    where = sub_cut(rank_tots)
    if where is None:
        return dC, dF, -1, -1
    r, need = where
    total = np.zeros(256, np.int64)
    for h in sub_hists:
        total += np.asarray(h, dtype=np.int64).reshape(256)
    return dC, dF, sub_from_hist(total), int(lo_of_rank[r]) + local_cut(lens_of_rank[r], need)


def agree_params(dist, mine, lens, lo, sub_hist_fn):
    """The same through torch.distributed (host tensors; gloo or any backend that moves them):
    `mine` = this rank's (delChar, del_first_global) from its own slice, `lens` its entry lengths,
    `lo` its first global entry, sub_hist_fn(i0, i1) -> 256 counts of the substitution-QV bytes of
    its local entries [i0, i1).  Two tiny collectives (world x 3 int64, then 256 + 1 int64); every
    rank returns the same (delChar, del_first, subChar, sub_first)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    lens = np.asarray(lens)
    tot = int(lens.astype(np.uint64).sum())
    every = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(every, torch.tensor([int(mine[0]), int(mine[1]), tot], dtype=torch.int64))
    dC, dF = merge_del([(int(e[0]), int(e[1])) for e in every])
    where = sub_cut([int(e[2]) for e in every])
    msg = torch.zeros(257, dtype=torch.int64)
    if where is not None:
        r, need = where
        if rank < r and len(lens):
            msg[:256] = torch.as_tensor(np.asarray(sub_hist_fn(0, len(lens)), dtype=np.int64))
        elif rank == r:
            cut = local_cut(lens, need)
            msg[:256] = torch.as_tensor(np.asarray(sub_hist_fn(0, cut + 1), dtype=np.int64))
            msg[256] = lo + cut
    dist.all_reduce(msg)
    if where is None:
        return dC, dF, -1, -1
    return dC, dF, sub_from_hist(msg[:256].numpy()), int(msg[256])


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
