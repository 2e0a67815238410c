"""N>1 path on CPU: two gloo ranks shard one .quiva file, exchange the scan state and histograms
on the host, build tables independently and produce record streams that concatenate to the
single-process file.  The per-shard compute here is the ORACLE standing in for the GPU kernels
(this test is about the sharding logic of dextractor_amd/shard.py, not about kernels)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
from dextractor_amd import _lib as L
from dextractor_amd import api, shard, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, text, off, lens, hdr4, lossy, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = len(lens)
        lo, hi = shard.entry_range(n, rank, world)
        # this rank's slice as its own .quiva image (entries lo..hi-1)
        start = int(off[lo]) - _hdr_len(text, int(off[lo])) if hi > lo else 0
        stop = int(off[hi - 1]) + 5 * (int(lens[hi - 1]) + 1) if hi > lo else 0
        piece = text[start:stop]
        st = O.qv_scan(piece) if hi > lo else None
        t8 = np.frombuffer(text, np.uint8)

        def sub_hist(i0, i1):                                 # substitution-QV bytes of local entries [i0, i1)
            h = np.zeros(256, np.int64)
            for i in range(lo + i0, lo + i1):
                Ln, o = int(lens[i]), int(off[i])
                h += np.bincount(t8[o + 4 * (Ln + 1): o + 4 * (Ln + 1) + Ln], minlength=256)
            return h

        mine = (st.delChar, lo + st.del_first) if st is not None and st.delChar >= 0 else (-1, -1)
        dC, dF, sC, sF = shard.agree_params(dist, mine, lens[lo:hi], lo, sub_hist)
        # raw histograms of the slice under the GLOBAL scan state
        h = _raw_hist_with_params(piece, lo, dC, dF, sC, sF)
        ht = torch.from_numpy(np.concatenate([h.reshape(-1), [int(lens[lo:hi].astype(np.uint64).sum())]]).astype(np.int64))
        dist.all_reduce(ht)                                   # host-side sum, 12 KB
        hist, tot = ht[:-1].numpy().astype(np.uint64).reshape(6, 256), int(ht[-1])
        coding = api.qv_build(hist, tot, L.QVParams(dC, sC, dF, sF), lossy)
        # encode the slice with the shared tables; framing continues the well chain
        ref = O.Coding()
        for s in range(6):
            ref.s[s].type = coding.s[s].type
            for k in range(256):
                ref.s[s].bits[k] = coding.s[s].bits[k]; ref.s[s].lens[k] = coding.s[s].lens[k]
        ref.delChar, ref.subChar = coding.delChar, coding.subChar
        blob, hoff, _ = api.frame_headers(hdr4[lo:hi], None, shard.previous_well(hdr4, lo))
        out = []
        for i in range(lo, hi):
            Ln, o = int(lens[i]), int(off[i])
            lines = np.stack([t8[o + k * (Ln + 1): o + k * (Ln + 1) + Ln] for k in range(5)])
            body, _ = O.qv_encode_entry(ref, lossy, lines)
            out.append(blob[int(hoff[i - lo]): int(hoff[i - lo + 1])].tobytes() + body)
        head = b"\xaa\x55" + api.qv_write_coding(coding, text[: text.index(b"/", 1)]) if rank == 0 else b""
        q.put((rank, head, b"".join(out)))
    finally:
        dist.destroy_process_group()


def _hdr_len(text, data_off):
    return data_off - (text.rfind(b"\n", 0, data_off - 1) + 1)


def _raw_hist_with_params(piece, lo, dC, dF, sC, sF):
    """Histogram_Seqs + Histogram_Runs of a slice given the file-global run chars / start entries."""
    off, ln, _, _ = api.index_quiva(piece)
    t8 = np.frombuffer(piece, np.uint8)
    h = np.zeros((6, 256), np.uint64)
    for i in range(len(ln)):
        Ln, o = int(ln[i]), int(off[i])
        for s, k in ((0, 0), (1, 2), (2, 3), (3, 4)):
            h[s] += np.bincount(t8[o + k * (Ln + 1): o + k * (Ln + 1) + Ln], minlength=256).astype(np.uint64)
        for s, k, rc, first in ((4, 0, dC, dF), (5, 4, sC, sF)):
            if rc >= 0 and lo + i >= first:
                line = t8[o + k * (Ln + 1): o + k * (Ln + 1) + Ln]
                nz = np.flatnonzero(line != rc)
                runs = np.diff(np.concatenate([[-1], nz])) - 1
                if Ln and line[-1] == rc:
                    runs = np.concatenate([runs, [Ln - 1 - (nz[-1] if len(nz) else -1)]])
                h[s] += np.bincount(np.minimum(runs, 255), minlength=256).astype(np.uint64)
    return h


def _run_sharded(c, world, lossy):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, c.text, c.off, c.len, c.hdr, lossy, q))
             for r in range(world)]
    for p in procs: p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs: p.join(30)
    assert all(p.exitcode == 0 for p in procs)
    return shard.concat(got[0][1], [g[2] for g in got])


@pytest.mark.parametrize("lossy", [False, True])
def test_two_rank_sharded_dexqv_equals_single(lossy):
    c = synth.make_quiva(31, seed=91, mean=9000)              # > 200000 symbols: both run schemes
    assert _run_sharded(c, 2, lossy) == O.dexqv(c.text, lossy)


def test_four_rank_cut_beyond_rank0():
    """~250 k symbols over 4 ranks: the 100000-symbol threshold (QV.c:1006) lies in rank 1's slice, so
    subChar must come from the histograms of ranks 0 and 1 together, not from rank 0 alone."""
    c = synth.make_quiva(28, seed=17, mean=9000)
    lo1, _ = shard.entry_range(28, 1, 4)
    assert int(c.len[:lo1].astype(np.uint64).sum()) < 100000 <= int(c.len.astype(np.uint64).sum())
    want = O.dexqv(c.text, False)
    assert O.qv_scan(c.text).subChar >= 0                     # the single-process scan does choose one
    assert _run_sharded(c, 4, False) == want


def test_eight_rank_sharded_dexqv_equals_single():
    """BASELINE configs[4]'s world size: eight ranks, contiguous entry ranges, the 100000-symbol cut in rank 2's slice,
    -- the concatenation is the single-process file."""
    c = synth.make_quiva(41, seed=23, mean=9000)
    cum = np.cumsum(c.len.astype(np.uint64))
    cut_rank = next(k for k in range(8) if int(cum[shard.entry_range(41, k, 8)[1] - 1]) >= 100000)
    assert cut_rank >= 1                                      # the cut lies beyond rank 0
    assert _run_sharded(c, 8, False) == O.dexqv(c.text, False)


def test_three_rank_short_file_has_no_subchar():
    c = synth.make_quiva(9, seed=5, mean=6000)                # < 100000 symbols in all: no subChar anywhere
    assert int(c.len.astype(np.uint64).sum()) < 100000
    assert _run_sharded(c, 3, False) == O.dexqv(c.text, False)


def test_entry_range_covers():
    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            r = [shard.entry_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_merge_params():
    assert shard.merge_params([(-1, -1, 63, 4), (50, 17, -1, -1), (50, 30, -1, -1)]) == (50, 17, 63, 4)
    assert shard.merge_params([(-1, -1, -1, -1)]) == (-1, -1, -1, -1)
    # cut located across ranks: 60 k + 60 k symbols, threshold inside rank 1 (its 2nd entry)
    h0, h1 = np.zeros(256, np.int64), np.zeros(256, np.int64)
    h0[63], h0[40], h1[40] = 10, 9, 5
    got = shard.merge_params([(-1, -1, -1, -1), (50, 7, -1, -1)], rank_tots=[60000, 60000], sub_hists=[h0, h1],
                             lo_of_rank=[0, 6], lens_of_rank=[[10000] * 6, [30000, 30000]])
    assert got == (50, 7, 40, 7)                              # 40: 14 > 63: 10; entry 6 + 1
    assert shard.sub_cut([10, 20]) is None
    assert shard.local_cut([5, 5, 5], 10) == 1 and shard.local_cut([5, 5, 5], 11) == 2
