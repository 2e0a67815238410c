#!/usr/bin/env python3
"""dexqv-encode throughput on MI355X (BASELINE.json metric: input GB/s, round-trip bit-exact).

One "step" = one complete dexqv encode of the resident synthetic .quiva batch:
    k_qv_prescan -> k_qv_hist -> (12 KB to host, Huffman tables, ~6 KB back) -> k_qv_sizes ->
    k_scan -> k_qv_encode
with the input image already in HBM when the timed region starts and the .dexqv record stream
left in HBM.  Workload at N=1: BASELINE.json configs[3] -- 1 M entries x 10 kb (5e10 stream
bytes); weak scaling for N>1 (every rank holds its own 1 M-entry slice of one corpus; the only
exchange is the 12 KB histogram sum + 32 B of scan state on the host side, via gloo -- no RCCL
on the data path).

Prints ONE JSON line (rank 0).  `value` = 5 * bases * steps / wall over all ranks, in GB/s.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--entries", type=int, default=1_000_000, help="entries per GPU")
    ap.add_argument("--mean", type=int, default=10_000)
    ap.add_argument("--dist", default="fixed", choices=["fixed", "lognormal"])
    ap.add_argument("--lossy", action="store_true")
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--cpu-sample-entries", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="timing experiments with ablated kernels")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--workload", default="dexqv", choices=["dexqv", "dexta", "dexar"],
                    help="dexqv = BASELINE metric (default); dexta/dexar = configs[1]/[2] (2-bit pack + unpack)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU for dexta/dexar")
    ap.add_argument("--twopass", action="store_true",
                    help="dexqv: dx_qv_sizes + dx_qv_encode instead of dx_qv_encode_onepass (scratch slots + compaction)")
    ap.add_argument("--no-verify", action="store_true",
                    help="dexqv: skip the (untimed) full-size on-device decode + compare after the timed steps")
    return ap.parse_args()


def trace(msg):
    """BENCH_TRACE=1: stage markers on stderr (to localise a stall on the GPU box)."""
    if os.environ.get("BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.workload != "dexqv":
        return pack2_main(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DEXGPU_BENCH_ONE_DEVICE"):        # testing the N>1 path on a 1-GPU box
        local = 0

    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(local)

    from dextractor_amd import _lib as L
    from dextractor_amd import api, synth

    ctx = api.Context(local)
    n = args.entries
    entry0 = rank * n
    movie = "m000_000"
    hlen = 1 + len(movie) + 1 + 8 + 1 + 7 + 1 + 7 + 6 + 3 + 1

    # ---- corpus slice of this rank, generated straight into HBM -----------------------------------
    lens = synth.lengths(entry0 + n, args.seed, args.dist, args.mean)[entry0:]
    hdr4 = synth.headers(n, args.seed, lens, entry0)
    rec_bytes = hlen + 5 * (lens.astype(np.uint64) + 1)
    off = (np.concatenate([[0], np.cumsum(rec_bytes)[:-1]]) + hlen).astype(np.uint64)
    text_bytes = int(rec_bytes.sum())
    bases = int(lens.astype(np.uint64).sum())
    prof = synth.pacbio_profile()

    d_text = torch.empty(text_bytes + 64, dtype=torch.uint8, device="cuda")
    t_off, t_len = torch.from_numpy(off.view(np.int64)).cuda(), torch.from_numpy(lens.view(np.int32)).cuda()
    t_hdr4 = torch.from_numpy(hdr4.reshape(-1)).cuda()
    t_lut = torch.from_numpy(prof.table().reshape(-1)).cuda()

    class Ptr:                                   # torch-owned device memory seen through the C-ABI
        def __init__(self, t): self.t, self.ptr = t, t.data_ptr()

    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p_text, p_off, p_len = Ptr(d_text), Ptr(t_off), Ptr(t_len)
    ctx.synth_quiva(args.seed, entry0, n, p_off, p_len, Ptr(t_hdr4), Ptr(t_lut), prof.del_run, movie, p_text)
    ctx.sync()
    batch = ctx.qv_batch(p_text, p_off, p_len, n, text_bytes=text_bytes + 64)

    # record framing bytes (host, O(records)): well deltas need the previous slice's last well
    lwell0 = 0 if entry0 == 0 else int(synth.headers(1, args.seed, lens[:1] * 0, entry0 - 1)[0, 0])
    blob, hoff, _ = api.frame_headers(hdr4, None, lwell0)
    p_hdr = Ptr(torch.from_numpy(blob.copy()).cuda())
    p_hoff = Ptr(torch.from_numpy(hoff.view(np.int64)).cuda())
    p_rec = Ptr(torch.empty(n + 1, dtype=torch.int64, device="cuda"))
    p_seg = Ptr(torch.empty(5 * n, dtype=torch.int32, device="cuda"))
    out_cap = int(2.2 * bases) + int(hoff[-1]) + 4096
    p_out = Ptr(torch.empty(out_cap, dtype=torch.uint8, device="cuda"))

    state = {}

    def step():
        p = ctx.qv_prescan(batch, entry0)
        if world > 1:                        # one file sharded over the ranks: agree on the scan state
            mine = torch.tensor([p.delChar, p.del_first, p.subChar, p.sub_first], dtype=torch.int64)
            every = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(every, mine)
            found = [e for e in every if e[0] >= 0]
            d = found[0] if found else every[0]
            p = L.QVParams(int(d[0]), int(every[0][2]), int(d[1]), int(every[0][3]))
        hist, tot = ctx.qv_hist(batch, p, entry0)
        if world > 1:                        # host-side sum of the 12 KB histograms (no RCCL)
            h = torch.from_numpy(np.concatenate([hist.reshape(-1), [tot]]).astype(np.int64))
            dist.all_reduce(h)
            hist, tot = h[:-1].numpy().astype(np.uint64).reshape(6, 256), int(h[-1])
        th = time.perf_counter()
        coding = api.qv_build(hist, tot, p, args.lossy)
        state["host_build_us"] = round((time.perf_counter() - th) * 1e6, 1)      # Huffman tables on the host
        ctx.qv_set_coding(coding, args.lossy)
        if not args.twopass:
            total = ctx.qv_encode_onepass(batch, p_hdr, p_hoff, p_seg, p_rec, p_out, out_cap)
        else:
            total = ctx.qv_sizes(batch, p_hoff, p_seg, p_rec)
            assert total <= out_cap, (total, out_cap)
            try:
                ctx.qv_encode(batch, p_hdr, p_hoff, p_rec, p_seg, p_out)
            except L.DexGPUError:
                if not args.no_check:
                    raise
        state.update(total=total, coding=coding, params=p)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    times = ctx.kernel_times()
    ctx.profile(False)

    roundtrip = None
    if not args.no_verify:
        # size-independent property at full size: decode(encode(x)) == x, entirely on the device
        # (the decoder is fed by the encoder's own index); headers are copied, data lines rebuilt
        trace("verify: allocate")
        d_back = torch.zeros_like(d_text)
        torch.cuda.synchronize()
        ctx.profile(True)
        trace("verify: decode")
        ctx.qv_decode(p_out, p_rec, p_hoff, p_seg, p_len, n, True, Ptr(d_back), p_off)
        ctx.sync(); torch.cuda.synchronize()
        trace("verify: compare")
        dec_ms = ctx.kernel_times().get("k_qv_decode", (0.0, 0))[0]
        ctx.profile(False)
        state["decode"] = {"kernel": "k_qv_decode + k_qv_decode_tags", "ms": round(dec_ms, 2),
                           "output_GBps": round(5.0 * bases / (dec_ms * 1e-3) / 1e9, 1) if dec_ms else None}
        # compare only the 5 data lines of every entry (fixed-length corpora: one strided view)
        if args.dist == "fixed":
            rec = hlen + 5 * (args.mean + 1)
            a = d_text[: n * rec].view(n, rec)[:, hlen:]
            b = d_back[: n * rec].view(n, rec)[:, hlen:]
            roundtrip = bool(torch.equal(a, b))
        else:
            idx = torch.from_numpy(np.concatenate([[0], np.cumsum(rec_bytes)]).astype(np.int64)).cuda()
            ok = True
            for i in range(0, n, max(1, n // 64)):           # sampled entries for ragged corpora
                lo, hi = int(off[i]), int(off[i]) + 5 * (int(lens[i]) + 1)
                ok = ok and bool(torch.equal(d_text[lo:hi], d_back[lo:hi]))
            roundtrip = ok
        del d_back

    # GPU text front end on the same resident image (untimed extra): newline scan -> entry index
    front = None
    if not args.no_verify:
        torch.cuda.synchronize()
        ctx.profile(True)
        trace("front end: index")
        t1 = time.perf_counter()
        o2, l2, h2, pl2 = ctx.index_quiva_device(p_text, text_bytes)
        trace("front end: done")
        t2 = time.perf_counter()
        kt = ctx.kernel_times().get("k_index", (0.0, 0))
        ctx.profile(False)
        front = {"entries": int(len(l2)), "wall_ms": round((t2 - t1) * 1e3, 2), "kernel_ms": round(kt[0], 3),
                 "text_GBps_kernels": round(text_bytes / (kt[0] * 1e-3) / 1e9, 1) if kt[0] else None,
                 "index_identical": bool((o2 == off).all() and (l2 == lens).all() and (h2 == hdr4).all())}

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        bb = torch.tensor([bases, state["total"]], dtype=torch.int64)
        dist.all_reduce(bb)
        all_bases, all_out = int(bb[0]), int(bb[1])
    else:
        all_bases, all_out = bases, state["total"]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    value = 5.0 * all_bases * args.steps / dt / 1e9

    # ---- roofline of the dominant kernel (per-launch algorithmic bytes / measured launch time) -----
    algo = {"k_qv_hist": 4.0 * bases, "k_qv_sizes": 4.0 * bases,
            "k_qv_encode": 5.0 * bases + state["total"]}
    kern = {k: {"ms_avg": ms / cnt, "launches": cnt} for k, (ms, cnt) in times.items()}
    for k, b in algo.items():
        if k in kern:
            per_step = max(1, round(kern[k]["launches"] / args.steps))   # the one-pass encoder runs in groups
            algo[k] = b / per_step
            kern[k]["launches_per_step"] = per_step
            kern[k]["algo_bytes"] = algo[k]
            kern[k]["GBps"] = algo[k] / (kern[k]["ms_avg"] * 1e-3) / 1e9
    dom = max(algo, key=lambda k: kern.get(k, {}).get("ms_avg", 0.0) * kern.get(k, {}).get("launches_per_step", 1))
    traffic = None
    if os.path.exists(args.traffic_file):
        try:
            tf = json.load(open(args.traffic_file))
            if tf.get("entries") == n and tf.get("mean") == args.mean and tf.get("dist") == args.dist:
                traffic = tf["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None and tf.get("launches_per_step", {}).get(dom, 1) != kern[dom]["launches_per_step"]:
                    traffic = None                           # profile taken with another grouping
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": round(kern[dom]["GBps"], 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(kern[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algo_bytes_per_launch": algo[dom]}
    # whole pipeline against the same roofline: (4 + 5) B/base read + output written, over the step's
    # wall time (kernels overlap in the one-pass encoder, so their durations do not add up)
    step_s = dt / args.steps
    pipe = {"algo_bytes": 9.0 * bases + state["total"], "step_ms": round(step_s * 1e3, 3),
            "kernel_ms_sum": round(sum(kern[k]["ms_avg"] * kern[k]["launches"] / args.steps for k in kern if k != "k_synth"), 3),
            "GBps": round((9.0 * bases + state["total"]) / step_s / 1e9, 1),
            "encoder": "two pass (sizes, encode)" if args.twopass else "one pass (scratch slots, compaction on a second stream)"}
    pipe["frac"] = round(pipe["GBps"] / HBM_PEAK_GBS, 4)

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(ctx, api, batch, d_text, off, lens, hdr4, hlen, args, state,
                           dict(p_out=p_out, p_rec=p_rec, n=n, movie=movie))

    line = {
        "metric": "dexqv encode input GB/s (5 QV/tag stream bytes per base; .dexqv bit-exact vs reference)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"dexqv 5-stream Huffman encode, {n} x {args.mean} .quiva per GPU "
                               f"({args.dist} lengths, BASELINE.json configs[3]), HBM-resident",
                   "entries_per_gpu": n, "mean_len": args.mean, "lossy": bool(args.lossy),
                   "input_bytes_per_gpu": 5 * bases, "text_image_bytes_per_gpu": text_bytes,
                   "output_bytes": all_out, "ratio": round(5.0 * all_bases / all_out, 3),
                   "sharding": "contiguous entry ranges, one file (host-side 12 KB histogram sum)" if world > 1 else "single GPU"},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "roundtrip_bit_exact": roundtrip,
        "host_table_build_us": state.get("host_build_us"),
        "decode": state.get("decode"),
        "text_front_end": front,
        "pipeline": pipe,
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in kern.items()},
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(ctx, api, batch, d_text, off, lens, hdr4, hlen, args, state, big):
    """Time the CPU path on a bounded sample of the same corpus (first S entries), on this host.
    kind "reference": the real reference `dexqv` (oracle/_ref, compiled from the reference's own
    sources); kind "port": the oracle's C restatement.  Also checks the GPU output for the same
    sample against it (bit-exact)."""
    S = min(args.cpu_sample_entries, len(lens))
    end = int(off[S - 1] + 5 * (int(lens[S - 1]) + 1)) if S else 0
    sample = d_text[:end].cpu().numpy().tobytes()
    sbases = int(lens[:S].astype(np.uint64).sum())
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "dexqv")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    res = {"unit": "GB/s", "cores": 1,
           "sample": f"first {S} entries of the bench corpus ({5 * sbases / 1e9:.2f} GB of stream bytes)"}
    want = None
    if os.path.isfile(ref_bin):
        with tempfile.TemporaryDirectory(dir=shm) as d:
            src = os.path.join(d, "s.quiva")
            with open(src, "wb") as f:
                f.write(sample)
            t0 = time.perf_counter()
            subprocess.check_call([ref_bin, "-k"] + (["-l"] if args.lossy else []) + [src])
            dt = time.perf_counter() - t0
            with open(os.path.join(d, "s.dexqv"), "rb") as f:
                want = f.read()
        res.update(kind="reference", value=round(5 * sbases / dt / 1e9, 4), seconds=round(dt, 2))
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        t0 = time.perf_counter()
        want = O.dexqv(sample, args.lossy)
        dt = time.perf_counter() - t0
        res.update(kind="port", value=round(5 * sbases / dt / 1e9, 4), seconds=round(dt, 2))
    trace("cpu_baseline: reference done; GPU file driver on the sample")
    got = ctx.dexqv(sample, args.lossy)          # same sample through the GPU path (file driver)
    trace("cpu_baseline: CLI end to end")
    res["gpu_output_identical"] = bool(got == want)

    # secondary number: end-to-end wall time of the drop-in CLI on the same sample file (tmpfs -> tmpfs,
    # includes process start, HIP init, PCIe both ways and the host parse) -- never the headline value
    cli = os.path.join(ROOT, "dextractor_amd", "bin", "dexqv")
    if os.path.isfile(cli):
        with tempfile.TemporaryDirectory(dir=shm) as d:
            src = os.path.join(d, "s.quiva")
            with open(src, "wb") as f:
                f.write(sample)
            t0 = time.perf_counter()
            rc_ = subprocess.call([cli, "-k"] + (["-l"] if args.lossy else []) + [src])
            dtc = time.perf_counter() - t0
            same = False
            if rc_ == 0:
                with open(os.path.join(d, "s.dexqv"), "rb") as f:
                    same = f.read() == want
        res["cli_end_to_end"] = {"seconds": round(dtc, 2), "GBps": round(5 * sbases / dtc / 1e9, 3),
                                 "output_identical": bool(same)}

    # how the single-threaded reference would be deployed: one independent copy per host core
    trace("cpu_baseline: all cores")
    cores = os.cpu_count() or 1
    if os.path.isfile(ref_bin) and cores > 1:
        S2 = min(4000, S)
        end2 = int(off[S2 - 1] + 5 * (int(lens[S2 - 1]) + 1))
        b2 = int(lens[:S2].astype(np.uint64).sum())
        with tempfile.TemporaryDirectory(dir=shm) as d:
            src = os.path.join(d, "s.quiva")
            with open(src, "wb") as f:
                f.write(sample[:end2])
            paths = []
            for k in range(cores):
                os.mkdir(os.path.join(d, f"c{k}"))
                os.symlink(src, os.path.join(d, f"c{k}", "s.quiva"))
                paths.append(os.path.join(d, f"c{k}", "s.quiva"))
            t0 = time.perf_counter()
            procs = [subprocess.Popen([ref_bin, "-k"] + (["-l"] if args.lossy else []) + [q]) for q in paths]
            rcs = [q.wait() for q in procs]
            dt = time.perf_counter() - t0
        if all(r == 0 for r in rcs):
            res["all_cores"] = {"cores": cores, "value": round(cores * 5 * b2 / dt / 1e9, 3), "unit": "GB/s",
                                "sample": f"{cores} concurrent copies of the reference, {S2} entries each"}

    # the big batch's LAST records (file offsets beyond 4 GiB) decoded by the oracle with the
    # batch's own tables must reproduce the last entries of the input image
    trace("cpu_baseline: tail records")
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        T, n = min(64, big["n"]), big["n"]
        rec = big["p_rec"].t[n - T: n + 1].cpu().numpy().astype(np.uint64)
        body = big["p_out"].t[int(rec[0]): int(rec[-1])].cpu().numpy().tobytes()
        img = b"\xaa\x55" + api.qv_write_coding(state["coding"], ("@" + big["movie"]).encode()) + body
        txt = O.undexqv(img, upper=True)
        lo = int(off[n - T])
        want_txt = d_text[lo - hlen: int(off[n - 1]) + 5 * (int(lens[n - 1]) + 1)].cpu().numpy().tobytes()
        strip = lambda b_: [ln for ln in b_.split(b"\n") if not ln.startswith(b"@")]
        res["tail_records_decode_ok"] = bool(strip(txt) == strip(want_txt)) if not args.lossy else None
    except Exception as e:                                   # the checker is optional plumbing
        res["tail_records_decode_ok"] = f"check failed to run: {e}"
    return res


def pack2_main(args):
    """BASELINE configs[1]/[2]: 2-bit pack + unpack of `--reads` x `--mean` reads (80-column text).
    A host-generated tile of 20000 reads is replicated on the device to the full size."""
    import torch
    from dextractor_amd import _lib as L
    from dextractor_amd import api, synth
    torch.cuda.set_device(0)
    ctx = api.Context(0)
    arrow = args.workload == "dexar"
    n0 = min(20000, args.reads)
    tile = synth.make_seqfile("arrow" if arrow else "fasta", n0, seed=args.seed, dist="fixed", mean=args.mean)
    reps = max(1, args.reads // n0)
    n = n0 * reps
    tb = len(tile.text)
    t_tile = torch.from_numpy(np.frombuffer(tile.text, np.uint8).copy()).cuda()
    d_text = t_tile.repeat(reps)
    rep_off = (np.arange(reps, dtype=np.uint64) * np.uint64(tb))[:, None]
    off = (rep_off + tile.off[None, :]).reshape(-1)
    tlen, nsym = np.tile(tile.tlen, reps), np.tile(tile.len, reps)
    hdr4 = np.tile(tile.hdr, (reps, 1)).astype(np.int32)
    hdr4[:, 0] = np.arange(1, n + 1)                              # wells keep increasing across the tiles
    cnr = None
    if arrow:
        cnr = np.tile((np.array(tile.snr, dtype=np.float64) * 100 + 0.5).astype(np.uint16), (reps, 1))
    blob, hoff, _ = api.frame_headers(hdr4, cnr)
    clen = (nsym.astype(np.uint64) + 3) >> 2
    rec = (hoff[1:] - hoff[:-1]) + clen
    ooff = np.concatenate([[0], np.cumsum(rec)[:-1]]).astype(np.uint64)
    out_bytes = int(rec.sum())

    class Ptr:
        def __init__(self, t): self.t, self.ptr = t, t.data_ptr()
    up = lambda a, dt: Ptr(torch.from_numpy(np.ascontiguousarray(a).view(dt)).cuda())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p_text, p_off, p_tlen, p_nsym = Ptr(d_text), up(off, np.int64), up(tlen, np.int32), up(nsym, np.int32)
    p_hdr, p_hoff, p_ooff = up(blob, np.uint8), up(hoff, np.int64), up(ooff, np.int64)
    p_out = Ptr(torch.empty(out_bytes + 64, dtype=torch.uint8, device="cuda"))
    # decode side: packed bytes sit after each record's framing; text goes back in 80-column lines
    ioff = (ooff + (hoff[1:] - hoff[:-1])).astype(np.uint64)
    p_ioff = up(ioff, np.int64)
    p_back = Ptr(torch.zeros(n * tb // n0 + 64, dtype=torch.uint8, device="cuda"))
    alpha = L.DX_ALPHA_ARROW if arrow else L.DX_ALPHA_BASES
    letters = L.DX_LETTERS_ARROW if arrow else L.DX_LETTERS_UPPER

    def step():
        ctx.pack2_encode(alpha, p_text, p_off, p_tlen, p_nsym, n, p_hdr, p_hoff, p_out, p_ooff)
        ctx.pack2_decode(letters, p_out, p_ioff, p_nsym, n, 80, p_back, p_off)

    for _ in range(args.warmup):
        step()
    ctx.sync(); torch.cuda.synchronize()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    times = ctx.kernel_times()
    ctx.profile(False)
    # round trip: the sequence text of every read of the first and the last tile must come back
    # byte-identical (header lines are not rewritten by the decoder)
    seqmask = np.zeros(tb, dtype=bool)
    for o_, t_ in zip(tile.off, tile.tlen):
        seqmask[int(o_): int(o_) + int(t_)] = True
    same = True
    for t in sorted({0, reps - 1}):
        a = d_text[t * tb: (t + 1) * tb].cpu().numpy()
        b = p_back.t[t * tb: (t + 1) * tb].cpu().numpy()
        same = same and bool(np.array_equal(a[seqmask], b[seqmask]))
    bases = int(nsym.astype(np.uint64).sum())
    text_in = int(tlen.astype(np.uint64).sum())
    enc_ms, dec_ms = times["k_pack2_encode"][0] / args.steps, times["k_pack2_decode"][0] / args.steps
    algo_enc, algo_dec = text_in + out_bytes, (out_bytes + text_in)
    line = {"metric": f"{args.workload} 2-bit pack input GB/s (+ unpack); round-trip bit-exact",
            "value": round(text_in / (enc_ms * 1e-3) / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload} + un{args.workload}, {n} x {args.mean} reads, 80-column text, HBM-resident "
                                   f"(BASELINE.json configs[{2 if arrow else 1}])", "reads": n, "bases": bases,
                       "text_bytes": text_in, "packed_bytes": out_bytes},
            "roofline": {"kernel": "k_pack2_encode", "bound": "hbm", "achieved": round(algo_enc / (enc_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_enc / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "algo_bytes_per_launch": algo_enc},
            "decode": {"kernel": "k_pack2_decode", "ms": round(dec_ms, 3), "GBps": round(algo_dec / (dec_ms * 1e-3) / 1e9, 1),
                       "frac": round(algo_dec / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "encode_ms": round(enc_ms, 3), "roundtrip_bit_exact": same, "cpu_baseline": None}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
