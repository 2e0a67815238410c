#!/usr/bin/env python3
"""dexqv-encode throughput on MI355X (BASELINE.json metric: input GB/s at 1/2/4/8 GPUs, round-trip bit-exact).

One "step" = one complete dexqv encode of the resident synthetic .quiva batch by the PRODUCT encoder
(the one dx_file_dexqv and the CLI use):
    k_qv_prescan -> k_qv_hist -> (12 KB to host, Huffman tables, ~6 KB back) -> dx_qv_encode_onepass
with the input image already in HBM when the timed region starts and the .dexqv record stream left in
HBM.  Workload at N=1: BASELINE.json configs[3] -- 1 M entries x 10 kb (5e10 stream bytes).  N>1:
configs[4] -- every rank holds a 2.5 M-entry slice (125 GB of stream bytes; 1 TB over 8 GPUs) of ONE
corpus, contiguous entry ranges; the only exchange is host-side (gloo): the scan state and the 12 KB
histogram sum, after which every rank builds identical tables -- no RCCL on the data path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` without a launcher starts the N ranks itself (child processes, before anything touches the
GPU); under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one rank.
Prints ONE JSON line (rank 0).  `value` = 5 * bases * steps / wall over all ranks, in GB/s.  At N=1 the
line also carries, under "other_workloads", the lognormal-length dexqv run and the dexta / dexar
(BASELINE configs[1]/[2]) runs with their own roofline and CPU baseline (skip: --only-main).
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ENTRIES_1GPU = 1_000_000         # BASELINE configs[3]
ENTRIES_SHARD = 2_500_000        # BASELINE configs[4]: 1 TB over 8 GPUs = 2.5 M entries x 10 kb per GPU


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--entries", type=int, default=None,
                    help=f"entries per GPU (default {ENTRIES_1GPU} at N=1, {ENTRIES_SHARD} per rank at N>1)")
    ap.add_argument("--mean", type=int, default=10_000)
    ap.add_argument("--dist", default="fixed", choices=["fixed", "lognormal", "mixed", "short_u", "mixed_tail"])
    ap.add_argument("--pipeline", action="store_true",
                    help="dexqv: the timed steps as consecutive batches of one job (dx_qv_encode_onepass_begin / _end: a step's "
                         "last compaction runs beside the next step's scan) instead of every step ending its own encode")
    ap.add_argument("--with-index", action="store_true",
                    help="dexqv: the timed steps also write the group index (dx_qv_subindex) -- what it costs the encoder")
    ap.add_argument("--lossy", action="store_true")
    ap.add_argument("--del-run-p", type=float, default=0.85, help="dexqv: share of the run character in the deletion QV line")
    ap.add_argument("--sub-run-p", type=float, default=0.80, help="dexqv: share of the run character in the substitution QV line")
    ap.add_argument("--seed", type=int, default=20261003)
    ap.add_argument("--cpu-sample-entries", type=int, default=20000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="timing experiments with ablated kernels")
    ap.add_argument("--traffic-file", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--workload", default="dexqv", choices=["dexqv", "dexta", "dexar"],
                    help="dexqv = BASELINE metric (default); dexta/dexar = configs[1]/[2] (2-bit pack + unpack)")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU for dexta/dexar")
    ap.add_argument("--p2-align", type=int, default=0,
                    help="dexta/dexar experiment: unpack every read's text to an offset that is a multiple of this (0: back into the "
                         "file image's own, arbitrarily aligned places; the round trip is then not checked)")
    ap.add_argument("--twopass", action="store_true",
                    help="dexqv: dx_qv_sizes + dx_qv_encode instead of dx_qv_encode_onepass (scratch slots + compaction)")
    ap.add_argument("--no-verify", action="store_true",
                    help="dexqv: skip the (untimed) full-size on-device decode + compare after the timed steps")
    ap.add_argument("--scratch-budget-gb", type=float, default=None,
                    help="dexqv: memory the one-pass encoder may take for its scratch regions (dx_set_scratch_budget); default: "
                         "none at N=1 (by free device memory), 64 at N>1 so that every rank of a sharded job takes the same route")
    ap.add_argument("--cli-large-gb", type=float, default=20.0,
                    help="size of the .quiva the tools are timed on end to end in the cpu_baseline leg (tools/cli_scale.py); 0: not at all")
    ap.add_argument("--no-walk-index", action="store_true",
                    help="dexqv: skip the decode of the batch with the HOST walk's group index (brings the 14 GB stream to the host and walks it)")
    ap.add_argument("--only-main", action="store_true",
                    help="N=1: only the headline workload, without the lognormal / dexta / dexar extras")
    return ap.parse_args()


def trace(msg):
    """BENCH_TRACE=1: stage markers on stderr (to localise a stall on the GPU box)."""
    if os.environ.get("BENCH_TRACE"):
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------
#  --gpus N without a launcher: start the N ranks as child processes BEFORE anything touches the GPU
# ------------------------------------------------------------------------------------------------
def launch_ranks(n):
    """One child process per GPU (never exec from a process that has initialised HIP: this parent has
    imported neither torch nor the library).  Rank 0 inherits stdout and prints the JSON line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, left = 0, set(range(n))
    while left:                                   # a failing rank takes the others down with it (exact PIDs)
        for r in sorted(left):
            c = procs[r].poll()
            if c is None:
                continue
            left.discard(r)
            if c != 0 and rc == 0:
                rc = c
                print(f"bench.py: rank {r} exited with {c}; stopping the other ranks", file=sys.stderr)
                for q in left:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc if rc >= 0 else 1


class Ptr:                                        # torch-owned device memory seen through the C-ABI
    def __init__(self, t):
        self.t, self.ptr = t, t.data_ptr()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (starts N "
              f"ranks itself) or `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bench.py "
              f"--gpus N`", file=sys.stderr)
        sys.exit(2)
    if args.workload != "dexqv":
        if world > 1:
            print("bench.py: the dexta/dexar lines are single-GPU (reads shard with no exchange at all)", file=sys.stderr)
            sys.exit(2)
        print(json.dumps(pack2_bench(args, args.workload == "dexar")))
        return
    if os.environ.get("DEXGPU_BENCH_ONE_DEVICE"):        # testing the N>1 path on a 1-GPU box
        local = 0
    if args.entries is None:
        args.entries = ENTRIES_1GPU if world == 1 else ENTRIES_SHARD

    line = dexqv_bench(args, rank, world, local, cpu=(world == 1 and not args.no_cpu_baseline))
    if rank != 0:
        return
    if world == 1 and not args.only_main:
        extra = {}
        a2 = argparse.Namespace(**vars(args))
        a2.dist = "lognormal" if args.dist == "fixed" else "fixed"
        trace("extra: dexqv, other length distribution")
        try:
            l2 = dexqv_bench(a2, 0, 1, local, cpu=False, front=False)
            extra["dexqv_" + a2.dist] = {k: l2[k] for k in ("value", "unit", "ms_per_step", "config", "roofline",
                                                              "roundtrip_bit_exact", "pipeline", "decode", "decode_indexed")}
        except Exception as e:                               # an extra must never cost the headline line
            extra["dexqv_" + a2.dist] = {"error": repr(e)}
        # run density: the tokens of the run-coded lines hold runs up to 126 and longer ones by exception, so the
        # encoder's route must not depend on how long the runs are (share of entries on the text-reading encoder reported)
        sweep = {}
        for pr in (0.5, 0.95, 0.99):
            a3 = argparse.Namespace(**vars(args))
            a3.del_run_p, a3.sub_run_p, a3.steps, a3.warmup = pr, pr, 3, 1
            a3.no_walk_index = True
            trace(f"extra: dexqv, run density {pr}")
            try:
                # (decode with the encoder's group index too: a pass of 512 tokens covers 10 k positions at 0.95 and 51 k at
                #  0.99 -- more than the decoder stages at a time --, and most lanes hold runs of 127 and more at 0.99)
                l3 = dexqv_bench(a3, 0, 1, local, cpu=False, front=False, index_decode=True)
                sweep[str(pr)] = {"value": l3["value"], "unit": "GB/s", "ms_per_step": l3["ms_per_step"], "ratio": l3["config"]["ratio"],
                                  "roundtrip_bit_exact": l3["roundtrip_bit_exact"],
                                  "decode_ms": (l3.get("decode") or {}).get("ms"),
                                  "decode_indexed_ms": (l3.get("decode_indexed") or {}).get("ms"),
                                  "decode_indexed_bit_exact": (l3.get("decode_indexed") or {}).get("bit_exact"),
                                  "entries_on_text_encoder": l3["encoder_route"].get("text_entries"),
                                  "vs_default_density": round(l3["value"] / line["value"], 3)}
            except Exception as e:
                sweep[str(pr)] = {"error": repr(e)}
        extra["dexqv_run_density"] = sweep
        # entry length: a wave works an entry, a step is 1 KiB of each of its lines -- short entries fill neither
        shapes = {}
        for mean, entries in ((2000, 2_000_000), (300, 4_000_000)):
            a5 = argparse.Namespace(**vars(args))
            a5.mean, a5.entries, a5.steps, a5.warmup, a5.no_walk_index = mean, entries, 3, 1, True
            trace(f"extra: dexqv, {entries} entries of {mean}")
            try:
                l5 = dexqv_bench(a5, 0, 1, local, cpu=False, front=False, index_decode=False)
                shapes[f"{entries}x{mean}"] = {"value": l5["value"], "unit": "GB/s", "ms_per_step": l5["ms_per_step"],
                                               "roundtrip_bit_exact": l5["roundtrip_bit_exact"],
                                               "kernels": {k: round(v["ms_avg"], 3) for k, v in l5["kernels"].items()},
                                               "decode_ms": (l5.get("decode") or {}).get("ms")}
            except Exception as e:
                shapes[f"{entries}x{mean}"] = {"error": repr(e)}
        extra["dexqv_entry_length"] = shapes
        # a batch of short entries with long ones among them (nine in ten U[100, 600], one in ten lognormal around 10 kb: three quarters
        # of the bytes in the long ones) -- the lanes take the short ones, the wave-per-entry kernels the long ones as a batch of their
        # own -- beside its two parts alone, and the length-weighted mean of their rates (what the mix could reach at best)
        mixed = {}
        for name, dist_, entries in (("mixed", "mixed", 2_000_000), ("short_part", "short_u", 1_800_000), ("long_part", "lognormal", 200_000)):
            a6 = argparse.Namespace(**vars(args))
            a6.dist, a6.mean, a6.entries, a6.steps, a6.warmup, a6.no_walk_index = dist_, 10_000, entries, 3, 1, True
            trace(f"extra: dexqv, {name}")
            try:
                l6 = dexqv_bench(a6, 0, 1, local, cpu=False, front=False, index_decode=False)
                mixed[name] = {"value": l6["value"], "unit": "GB/s", "ms_per_step": l6["ms_per_step"], "roundtrip_bit_exact": l6["roundtrip_bit_exact"],
                               "GB_per_step": round(l6["value"] * l6["ms_per_step"] / 1e3, 3), "route": l6["encoder_route"].get("direct"),
                               "kernels": {k: round(v["ms_avg"], 3) for k, v in l6["kernels"].items()}}
            except Exception as e:
                mixed[name] = {"error": repr(e)}
        if all("value" in v for v in mixed.values()):
            gs, gl = mixed["short_part"]["GB_per_step"], mixed["long_part"]["GB_per_step"]
            best = (gs + gl) / (gs / mixed["short_part"]["value"] + gl / mixed["long_part"]["value"])
            mixed["weighted_mean_of_the_parts_GBps"] = round(best, 1)
            mixed["mixed_over_that"] = round(mixed["mixed"]["value"] / best, 3)
        extra["dexqv_mixed_lengths"] = mixed
        # BASELINE configs[4]: the slice ONE GPU of the 8-GPU job holds (2.5 M entries, 125 GB of QV bytes), under the scratch
        # budget every rank of that job runs with -- so that the per-GPU work of configs[4] is timed wherever this line is
        if args.entries == ENTRIES_1GPU and args.mean == 10_000:
            a4 = argparse.Namespace(**vars(args))
            a4.entries, a4.steps, a4.warmup, a4.no_walk_index = ENTRIES_SHARD, 3, 1, True
            a4.scratch_budget_gb = 64.0 if args.scratch_budget_gb is None else args.scratch_budget_gb
            trace("extra: dexqv, the configs[4] slice of one GPU")
            try:
                import torch
                torch.cuda.empty_cache()
                free_b = torch.cuda.mem_get_info()[0]
                if free_b < 262e9:
                    extra["config4_slice"] = {"skipped": f"{free_b / 1e9:.0f} GB of device memory free, the slice wants 262"}
                else:
                    l4 = dexqv_bench(a4, 0, 1, local, cpu=False, front=False, index_decode=False)
                    extra["config4_slice"] = {k: l4[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "roundtrip_bit_exact",
                                                                   "encoder_route", "kernels")}
            except Exception as e:
                extra["config4_slice"] = {"error": repr(e)}
        for w in ("dexta", "dexar"):
            trace(f"extra: {w}")
            try:
                extra[w] = pack2_bench(args, w == "dexar")
            except Exception as e:
                extra[w] = {"error": repr(e)}
        line["other_workloads"] = extra
    if rank == 0:
        # what a reader of the line's top level (scalars only) should see of the secondary figures
        def dig(d, *ks):
            for k in ks:
                d = d.get(k) if isinstance(d, dict) else None
            return d if isinstance(d, (int, float, bool)) else None
        flat = {"walk_and_decode_ms": dig(line, "device_walk", "walk_and_decode_ms"), "walk_kernel_ms": dig(line, "device_walk", "kernel_ms"),
                "walk_decode_bit_exact": dig(line, "device_walk", "decode_bit_exact"),
                "decode_indexed_ms": dig(line, "decode_indexed", "ms"),
                "lognormal_GBps": dig(line, "other_workloads", "dexqv_lognormal", "value"),
                "entries_2Mx2000_GBps": dig(line, "other_workloads", "dexqv_entry_length", "2000000x2000", "value"),
                "entries_4Mx300_GBps": dig(line, "other_workloads", "dexqv_entry_length", "4000000x300", "value"),
                "mixed_lengths_GBps": dig(line, "other_workloads", "dexqv_mixed_lengths", "mixed", "value"),
                "mixed_over_weighted_mean_of_parts": dig(line, "other_workloads", "dexqv_mixed_lengths", "mixed_over_that"),
                "run_density_0.5_GBps": dig(line, "other_workloads", "dexqv_run_density", "0.5", "value"),
                "run_density_0.99_GBps": dig(line, "other_workloads", "dexqv_run_density", "0.99", "value"),
                "config4_slice_GBps": dig(line, "other_workloads", "config4_slice", "value"),
                "dexta_encode_frac": dig(line, "other_workloads", "dexta", "roofline", "frac"),
                "dexta_decode_frac": dig(line, "other_workloads", "dexta", "decode", "frac"),
                "cli_dexqv_20GB_s": dig(line, "cpu_baseline", "cli_end_to_end_large", "dexqv", "s"),
                "cli_undexqv_20GB_s": dig(line, "cpu_baseline", "cli_end_to_end_large", "undexqv", "s")}
        for k, v in flat.items():
            if v is not None:
                line["x_" + k] = v
    print(json.dumps(line))


# ------------------------------------------------------------------------------------------------
#  dexqv (BASELINE configs[3] at N=1, configs[4] at N>1)
# ------------------------------------------------------------------------------------------------
def dexqv_bench(args, rank, world, local, cpu=True, front=True, index_decode=True):
    import torch
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # (gloo announces its connections on the C-level stdout -- "[Gloo] Rank 0 is connected to ..." --: stdout is for the
        #  ONE JSON line, so while the group forms, file descriptor 1 is pointed at stderr)
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(keep, 1)
            os.close(keep)
    if os.environ.get("DEXGPU_BENCH_ONE_DEVICE"):      # (testing the N > 1 path on a one-GPU box: every rank on device 0)
        local = 0
    torch.cuda.set_device(local)

    from dextractor_amd import _lib as L
    from dextractor_amd import api, shard, synth

    ctx = api.Context(local)
    budget_gb = args.scratch_budget_gb if args.scratch_budget_gb is not None else (64.0 if world > 1 else 0.0)
    if budget_gb:
        ctx.set_scratch_budget(int(budget_gb * 1e9))
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
    prof = synth.pacbio_profile(args.del_run_p, args.sub_run_p)

    d_text = torch.empty(text_bytes + 64, dtype=torch.uint8, device="cuda")
    t_off, t_len = torch.from_numpy(off.view(np.int64)).cuda(), torch.from_numpy(lens.view(np.int32)).cuda()
    t_hdr4 = torch.from_numpy(hdr4.reshape(-1)).cuda()
    t_lut = torch.from_numpy(prof.table().reshape(-1)).cuda()

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
    state = {"out_cap": 0, "p_out": None}

    def sub_hist(i0, i1):                        # substitution-QV bytes of local entries [i0, i1) (only when the
        h = np.zeros(256, np.int64)              # 100000-symbol threshold lies beyond rank 0: tiny slices)
        for i in range(i0, i1):
            o, ln = int(off[i]), int(lens[i])
            h += torch.bincount(d_text[o + 4 * (ln + 1): o + 4 * (ln + 1) + ln].long(), minlength=256).cpu().numpy()
        return h

    def step():
        if world > 1:                        # one file sharded over the ranks: agree on the scan state (host, gloo)
            p = ctx.qv_prescan(batch, entry0)
            dC, dF, sC, sF = shard.agree_params(dist, (p.delChar, p.del_first), lens, entry0, sub_hist)
            p = L.QVParams(dC, sC, dF, sF)
            mine, tot = ctx.qv_hist(batch, p, entry0)
        else:                                # QVcoding_Scan in one call: the scan state stays on the device (dx_qv_scan)
            p, mine, tot = ctx.qv_scan(batch, entry0)
        state["hist_mine"] = mine
        hist = mine
        if world > 1:                        # host-side sum of the 12 KB histograms (no RCCL)
            h = torch.from_numpy(np.concatenate([mine.reshape(-1), [tot]]).astype(np.int64))
            dist.all_reduce(h)
            hist, tot = h[:-1].numpy().astype(np.uint64).reshape(6, 256), int(h[-1])
        th = time.perf_counter()
        coding = api.qv_build(hist, tot, p, args.lossy)
        state["host_build_us"] = round((time.perf_counter() - th) * 1e6, 1)      # Huffman tables on the host
        ctx.qv_set_coding(coding, args.lossy)
        need = int(hoff[-1]) + api.qv_out_bound(mine, n, coding, args.lossy)     # from this rank's own counts
        if need > state["out_cap"]:          # first step only (the corpus does not change between steps)
            if state.get("begun"):
                state["total"] = ctx.qv_encode_onepass_end()
                state["begun"] = False
            state["p_out"] = None
            state["p_out"] = Ptr(torch.empty(need + 4096, dtype=torch.uint8, device="cuda"))
            state["out_cap"] = need + 4096
        p_out, out_cap = state["p_out"], state["out_cap"]
        if not args.twopass and args.pipeline:
            # consecutive steps are consecutive batches of one job: the encode is queued (begin) and collected (end) just
            # before the next one starts, so that the last group's compaction runs beside the next batch's scan
            if state.get("begun"):
                state["total"] = ctx.qv_encode_onepass_end()
            ctx.qv_encode_onepass_begin(batch, p_hdr, p_hoff, p_seg, p_rec, p_out, out_cap)
            state["begun"] = True
            total = state.get("total", 0)
        elif not args.twopass:
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
        if not args.twopass:
            state["route"] = ctx.qv_onepass_info()
            if not state.get("timing"):          # (a query of the driver: the warm-up steps', never the timed ones' -- 0.1 ms of a step's host time)
                state["mem_used_peak"] = max(state.get("mem_used_peak", 0), (lambda f, t: t - f)(*torch.cuda.mem_get_info()))

    def fence():
        if state.get("begun"):                # the job's last encode: its compaction belongs inside the timed region
            state["total"] = ctx.qv_encode_onepass_end()
            state["begun"] = False
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if args.with_index:
        ctx.qv_subindex(True)
    for _ in range(args.warmup):
        step()
    if state["p_out"] is None:               # --warmup 0: the output buffer is sized (untimed) before the clock starts
        step()
    fence()
    ctx.profile(True)
    state["timing"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    state["timing"] = False
    times = ctx.kernel_times()
    ctx.profile(False)
    ctx.qv_subindex(False)
    p_out = state["p_out"]

    roundtrip = None
    if not args.no_verify:
        # size-independent property at full size: decode(encode(x)) == x, entirely on the device (the decoder
        # is fed by the encoder's own index), in chunks of entries so that any corpus size fits.  The decoder
        # writes only the five data lines of an entry; the chunk buffer is zero elsewhere and header bytes are
        # never zero, so the number of differing bytes must equal the number of header bytes exactly.
        trace("verify: decode + compare")
        if args.lossy:                                            # what a lossy stream decodes back to: the text with QV.c:1355-1372's
            ctx.qv_lossy_text(batch)                              # rounding applied (in place: the timed steps are over, and the
            ctx.sync()                                            # rounding is idempotent -- every later encode of it gives the same stream)
        ends = np.concatenate([off[1:] - hlen, [text_bytes]]).astype(np.uint64)   # end of each entry's record
        dec_parts = {}

        class _At:                                                # a device address as qv_decode takes it
            def __init__(self, ptr): self.ptr = ptr

        def decode_all(idx=None):                                 # idx: the device walk's index instead of the encoder's own arrays
          roundtrip, dec_ms = True, 0.0
          dec_parts.clear()
          ctx.trim(1 | 2)                                         # the encoder's scratch and tokens go back to the device first
          torch.cuda.empty_cache()                                # as few decode launches as the free memory allows
          chunk = max(1, min(n, int(max(1e9, 0.5 * torch.cuda.mem_get_info()[0] - 4e9) // (5 * (args.mean + 1) + hlen))))   # (a launch = one pool of tasks)
          for a in range(0, n, chunk):
              b = min(n, a + chunk)
              lo, hi = int(off[a]) - hlen, int(ends[b - 1])
              d_back = torch.zeros(hi - lo + 64, dtype=torch.uint8, device="cuda")
              o_rel = Ptr(torch.from_numpy((off[a:b] - np.uint64(lo)).view(np.int64)).cuda())
              torch.cuda.synchronize()
              ctx.profile(True)
              if idx is None:
                  ctx.qv_decode(p_out, Ptr(p_rec.t[a:]), Ptr(p_hoff.t[a:]), Ptr(p_seg.t[5 * a:]), Ptr(t_len[a:]), b - a, True,
                                Ptr(d_back), o_rel)
              else:
                  ctx.qv_decode(p_out, _At(idx.rec_off.ptr + 8 * a), _At(idx.hdr_off.ptr + 8 * a), _At(idx.seg.ptr + 20 * a), _At(idx.len.ptr + 4 * a),
                                b - a, True, Ptr(d_back), o_rel)
              ctx.sync(); torch.cuda.synchronize()
              kt = ctx.kernel_times()
              dec_ms += sum(kt.get(k, (0.0, 0))[0] for k in L.DECODE_KERNELS)
              for k in L.DECODE_KERNELS:
                  if k in kt:
                      dec_parts[k] = round(dec_parts.get(k, 0.0) + kt[k][0], 3)
              ctx.profile(False)
              diff = 0
              for c0 in range(0, hi - lo, 1 << 28):                 # compared in slices: no chunk-sized temporaries
                  c1 = min(hi - lo, c0 + (1 << 28))
                  diff += int(torch.count_nonzero(d_back[c0:c1] != d_text[lo + c0: lo + c1]))
              roundtrip = roundtrip and diff == (b - a) * hlen
              del d_back, o_rel
          return roundtrip, dec_ms

        roundtrip, dec_ms = decode_all()                          # lane-per-line kernels (what a bare .dexqv gets too)
        state["decode"] = {"kernel": "k_qv_decode_plain + k_qv_decode + k_qv_decode_tags", "ms": round(dec_ms, 2), "ms_by_kernel": dict(dec_parts),
                           "output_GBps": round(5.0 * bases / (dec_ms * 1e-3) / 1e9, 1) if dec_ms else None,
                           "frac_of_hbm_peak": round((5.0 * bases + float(state["total"])) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec_ms else None}
        if not args.twopass and index_decode:
            # ... and with the encoder's group index (dx_qv_subindex): one more, untimed, encode that leaves the index,
            # then the plain lines are decoded a wavefront per line (k_qv_decode_sub)
            try:
                trace("verify: decode with the group index")
                ctx.qv_subindex(True)
                ctx.profile(True)
                step()
                fence()
                kt = ctx.kernel_times()
                enc_ix_ms = sum(kt.get(k, (0.0, 0))[0] for k in ("k_qv_prescan", "k_qv_hist", "k_qv_encode", "k_qv_compact", "k_scan", "k_qv_sizes"))
                ctx.profile(False)
                ok2, dec2_ms = decode_all()
                ctx.qv_subindex(False)
                roundtrip = roundtrip and ok2
                L64 = lens.astype(np.int64)
                words_ = 4 * ((((L64 + 15) >> 4) + 3) >> 2) + 3   # plain lines: a byte per 16 symbols; + a word per 8 tokens of the run-coded lines
                state["decode_indexed"] = {"kernel": "k_qv_decode_sub + k_qv_decode_runs + k_qv_decode + k_qv_decode_tags", "ms": round(dec2_ms, 2), "ms_by_kernel": dict(dec_parts),
                                           "frac_of_hbm_peak": round((5.0 * bases + float(state["total"])) / (dec2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec2_ms else None,
                                           "output_GBps": round(5.0 * bases / (dec2_ms * 1e-3) / 1e9, 1) if dec2_ms else None,
                                           "index_bytes_without_run_groups": int(4 * words_.sum()),
                                           "bit_exact": bool(ok2),                           # (lossy: against the text with the reference's rounding applied)
                                           "note": "index written by one extra untimed step of the same encoder; kernels of that step: "
                                                   + str(round(enc_ix_ms, 2)) + " ms"}

            except Exception as e:                                # (e.g. no memory left for the index beside a 125 GB shard)
                ctx.qv_subindex(False)
                ctx.profile(False)
                torch.cuda.empty_cache()
                state["decode_indexed"] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
        if not args.twopass and world == 1:
            # ... and as a bare file's record stream is indexed where it lies: dx_qv_walk_device, lanes taking pieces of 28 KiB
            # (no download, no host walk); its index against the encoder's own
            try:
                trace("verify: the record walk on the device")
                ctx.profile(True)
                t_w = time.perf_counter()
                dix = ctx.qv_walk_device(p_out, int(state["total"]), 0, state["coding"], 1, 0)
                torch.cuda.synchronize()
                t_w1 = time.perf_counter()
                kt = ctx.kernel_times().get("k_qv_walk")
                ctx.profile(False)
                same = dix.n == n
                if same:
                    for mine_, theirs_, nb_ in ((dix.seg, p_seg, 20 * n), (dix.rec_off, p_rec, 8 * (n + 1))):
                        a_ = np.zeros(nb_, np.uint8); b_ = np.zeros(nb_, np.uint8)
                        ctx._chk(ctx.lib.dx_d2h(ctx.h, a_.ctypes.data, mine_.ptr, nb_)); ctx._chk(ctx.lib.dx_d2h(ctx.h, b_.ctypes.data, theirs_.ptr, nb_))
                        same = same and bool((a_ == b_).all())
                state["device_walk"] = {"kernel": "k_walk_find + k_walk_pieces + k_walk_gather + k_walk_rooms + k_walk_index; decode: k_qv_decode_sync (timed as k_qv_decode_sub) + k_qv_decode_runs", "wall_ms": round((t_w1 - t_w) * 1e3, 2),
                                        "kernel_ms": round(kt[0], 2) if kt else None, "launches": kt[1] if kt else None,
                                        "pieces": dix.pieces, "piece_bytes": dix.piece_bytes, "records": dix.n,
                                        "index_identical_to_the_encoders": bool(same),
                                        "stream_GBps_wall": round(float(state["total"]) / (t_w1 - t_w) / 1e9, 1)}
                roundtrip = roundtrip and same
                if same:                                          # ... and the stream decoded from THAT index: a bare stream, device only
                    okw0, decw0_ms = decode_all(dix)              # (record and segment starts alone: the lane-per-line kernels)
                    parts0 = dict(dec_parts)
                    dix.use(p_out)                                # ... and with the run-coded lines' groups the walk has noted on its way
                    okw, decw_ms = decode_all(dix)
                    dix.use(None)
                    state["device_walk"].update({"decode_ms_from_this_index": round(decw_ms, 2), "decode_bit_exact": bool(okw and okw0), "ms_by_kernel": dict(dec_parts),
                                                 "walk_and_decode_ms": round((kt[0] if kt else 0.0) + decw_ms, 2),
                                                 "run_lines_without_groups": dix.gidx_none, "plain_lines_without_words": dix.gidx_nosync, "group_index_bytes": 4 * dix.gidx_words,
                                                 "decode_ms_without_the_groups": round(decw0_ms, 2), "ms_by_kernel_without_the_groups": parts0})
                    roundtrip = roundtrip and okw and okw0
                dix.free()
            except Exception as e:                                # (the bench's own, undamaged stream: a walk that turns it down, or a
                ctx.profile(False)                                # decode that fails, is a defect -- never "skipped"; no room for the
                if getattr(e, "code", None) == -6 or isinstance(e, torch.OutOfMemoryError):      # walk's scratch beside a 125 GB slice is not)
                    torch.cuda.empty_cache()
                    state["device_walk"] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
                else:
                    state["device_walk"] = {"failed": f"{type(e).__name__}: {e}"[:200], "index_identical_to_the_encoders": False, "decode_bit_exact": False}
                    roundtrip = False
        if not args.twopass and index_decode and not args.no_walk_index and world == 1:
            # ... and as a bare file gets it: the record stream goes to the host as a .dexqv image, dx_qv_walk_indexed finds the
            # segment boundaries AND leaves the group index (host threads; it passes every code anyway), dx_qv_use_index hands
            # it to the decoder: the wave-per-line kernels on a stream no encoder of this context has seen
            try:
                trace("verify: decode with the host walk's group index")
                t_w = time.perf_counter()
                head = b"\xaa\x55" + api.qv_write_coding(state["coding"], ("@" + movie).encode())
                img = np.empty(len(head) + int(state["total"]), np.uint8)
                img[: len(head)] = np.frombuffer(head, np.uint8)
                torch.from_numpy(img[len(head):]).copy_(p_out.t[: int(state["total"])])
                t_w1 = time.perf_counter()
                w = api.qv_walk(img, index=True)
                t_w2 = time.perf_counter()
                del img
                assert int(w["n"]) == n and (w["seg"].reshape(-1) == p_seg.t[: 5 * n].cpu().numpy().view(np.uint32)).all()
                d_gidx = torch.from_numpy(w["gidx"].view(np.int32)).cuda()
                d_goff = torch.from_numpy(w["gidx_off"].view(np.int64)).cuda()
                ctx.qv_use_index(p_out, p_seg, n, Ptr(d_gidx), Ptr(d_goff), w["gidx_none"])
                ok3, dec3_ms = decode_all()
                ctx.qv_use_index(None, None, 0, None, None)
                roundtrip = roundtrip and ok3
                state["decode_walk_indexed"] = {"kernel": "k_qv_decode_sub + k_qv_decode_runs + k_qv_decode + k_qv_decode_tags", "ms": round(dec3_ms, 2),
                                                "ms_by_kernel": dict(dec_parts), "bit_exact": bool(ok3),
                                                "frac_of_hbm_peak": round((5.0 * bases + float(state["total"])) / (dec3_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if dec3_ms else None,
                                                "output_GBps": round(5.0 * bases / (dec3_ms * 1e-3) / 1e9, 1) if dec3_ms else None,
                                                "index_bytes": int(4 * len(w["gidx"])), "lines_without_index": int(w["gidx_none"]),
                                                "host_walk_s": round(t_w2 - t_w1, 2), "download_s": round(t_w1 - t_w, 2),
                                                "note": "group index made by the host walk of the bare stream (dx_qv_walk_indexed), not by the encoder"}
                del d_gidx, d_goff, w
            except Exception as e:
                try:
                    ctx.qv_use_index(None, None, 0, None, None)
                except Exception:
                    pass
                ctx.profile(False)
                torch.cuda.empty_cache()
                state["decode_walk_indexed"] = {"skipped": f"{type(e).__name__}: {e}"[:200]}
    # GPU text front end on the same resident image (untimed extra): newline scan -> entry index
    fr = None
    if front and not args.no_verify:
        torch.cuda.synchronize()
        ctx.profile(True)
        trace("front end: index")
        t1 = time.perf_counter()
        o2, l2, h2, pl2 = ctx.index_quiva_device(p_text, text_bytes)
        t2 = time.perf_counter()
        kt = ctx.kernel_times().get("k_index", (0.0, 0))
        ctx.profile(False)
        fr = {"entries": int(len(l2)), "wall_ms": round((t2 - t1) * 1e3, 2), "kernel_ms": round(kt[0], 3),
              "text_GBps_kernels": round(text_bytes / (kt[0] * 1e-3) / 1e9, 1) if kt[0] else None,
              "index_identical": bool((o2 == off).all() and (l2 == lens).all() and (h2 == hdr4).all())}

    tables_same = None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        bb = torch.tensor([bases, state["total"], 1 if roundtrip in (True, None) else 0], dtype=torch.int64)
        dist.all_reduce(bb)
        all_bases, all_out = int(bb[0]), int(bb[1])
        if roundtrip is not None:
            roundtrip = int(bb[2]) == world
        # every rank must have built the same tables: compare the coding images
        import hashlib
        img = api.qv_write_coding(state["coding"], b"@" + movie.encode())
        sig = torch.tensor([int.from_bytes(hashlib.sha256(img).digest()[:7], "little")], dtype=torch.int64)
        every = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(every, sig)
        tables_same = all(int(e[0]) == int(every[0][0]) for e in every)
    else:
        all_bases, all_out = bases, state["total"]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        ctx.close()
        return None

    value = 5.0 * all_bases * args.steps / dt / 1e9

    # ---- roofline: the WHOLE STEP against HBM -- (4 B hist + 5 B encode) per base read + the output written,
    # over the step's wall time -- with every kernel of the step beside it: its share of the 8(d) bytes, the bytes it
    # must move IN THIS DESIGN (hist: 5 lines read + tokens written; fast encoder: tokens + plain lines read + output
    # written; compaction: nothing -- it is pure overhead), its measured time and its PMC traffic (profiles/traffic.json).
    step_s = dt / args.steps
    out_b = float(state["total"])
    cd, prm = state["coding"], state["params"]
    hist_mine = state.get("hist_mine")
    tok = 0.0                                              # tokens of the run-coded lines (2 bytes each, written and read once)
    if hist_mine is not None and not args.twopass and "no_tokens" not in (os.environ.get("DEXGPU_TEST") or ""):
        if prm.delChar >= 0:
            tok += float(hist_mine[0].sum() - hist_mine[0][prm.delChar])
        if prm.subChar >= 0:
            tok += float(hist_mine[3].sum() - hist_mine[3][prm.subChar])
    by_hist = (state.get("route") or {}).get("direct") == 2     # sizes from the entries' own histograms, records written in place
    tok_lines = (1 if prm.delChar >= 0 else 0) + (1 if cd.subChar >= 0 else 0)           # lines the encoder takes from tokens
    text_lines_enc = 5 - (2 if prm.delChar >= 0 else 0) - (1 if cd.subChar >= 0 else 0) if tok else 5   # (del tokens carry the tags)
    spec = {   # kernel id -> (8(d) bytes per step, design bytes per step)
        "k_qv_hist": (4.0 * bases, (5.0 if tok and prm.delChar >= 0 else 4.0) * bases + 2.0 * tok + (1536.0 * n if by_hist else 0.0)),
        "k_qv_encode": (5.0 * bases + out_b, 2.0 * tok + text_lines_enc * bases + out_b),
        "k_qv_encode_text": ((5.0 * bases + out_b) if args.twopass or not tok else 0.0, (5.0 * bases + out_b) if args.twopass or not tok else 0.0),
        "k_qv_sizes": (0.0, 1536.0 * n if by_hist else 0.0), "k_qv_compact": (0.0, 0.0), "k_scan": (0.0, 0.0), "k_qv_prescan": (0.0, 0.0),
    }
    if not tok and not args.twopass:                       # no tokens: the text-reading kernel is the encoder
        spec["k_qv_encode"] = (0.0, 0.0)
    kern, per_kernel = {}, {}
    for k, (ms, cnt) in times.items():
        kern[k] = {"ms_avg": ms / cnt, "launches": cnt}
        if k not in spec:
            continue
        lps = max(1, round(cnt / args.steps))
        ms_step = ms / args.steps
        a8d, design = spec[k]
        tr = read_traffic(args.traffic_file, "dexqv", k, dict(entries=n, mean=args.mean, dist=args.dist), lps, per_step=True)
        per_kernel[k] = {"ms_per_step": round(ms_step, 3), "launches_per_step": lps, "ms_avg_launch": round(ms / cnt, 4),
                         "algo_bytes_8d_per_step": a8d, "design_bytes_per_step": design, "traffic_bytes_per_step": tr,
                         "design_GBps": round(design / (ms_step * 1e-3) / 1e9, 1) if ms_step and design else None,
                         "traffic_GBps": round(tr / (ms_step * 1e-3) / 1e9, 1) if ms_step and tr else None}
    step_algo = 9.0 * bases + out_b
    step_traffic = None
    if per_kernel and all(v["traffic_bytes_per_step"] is not None for k, v in per_kernel.items() if spec[k][1] or k == "k_qv_compact"):
        step_traffic = sum(v["traffic_bytes_per_step"] or 0.0 for v in per_kernel.values())
    dom = max((k for k in per_kernel if spec[k][0]), key=lambda k: per_kernel[k]["ms_per_step"])
    dk = per_kernel[dom]
    dom_ms_launch = dk["ms_per_step"] / dk["launches_per_step"]
    roofline = {"kernel": "whole step (k_qv_prescan, k_qv_hist, host tables, k_qv_sizes_hist, k_qv_encode_fast in place)" if by_hist else
                          "whole step (k_qv_prescan, k_qv_hist, host tables, k_qv_encode_fast + k_qv_compact per group)",
                "bound": "hbm", "achieved": round(step_algo / step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(step_algo / step_s / 1e9 / HBM_PEAK_GBS, 4), "traffic": step_traffic,
                "traffic_source": "PMC passes of the committed evidence set (" + os.path.relpath(args.traffic_file, ROOT) + "): a constant of the tree, not a measurement of this run" if step_traffic is not None else None,
                "algo_bytes_per_step": step_algo,
                "formula": "SURVEY 8(d): (4 B/base histogram pass + 5 B/base encode pass + output bytes) / step wall time / 8 TB/s",
                "dominant_kernel": {"kernel": dom, "ms_avg_launch": round(dom_ms_launch, 4),
                                    "algo_bytes_8d_per_launch": dk["algo_bytes_8d_per_step"] / dk["launches_per_step"],
                                    "achieved": round(dk["algo_bytes_8d_per_step"] / (dk["ms_per_step"] * 1e-3) / 1e9, 1),
                                    "frac": round(dk["algo_bytes_8d_per_step"] / (dk["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "traffic": (dk["traffic_bytes_per_step"] / dk["launches_per_step"]) if dk["traffic_bytes_per_step"] else None},
                "per_kernel": per_kernel}
    if by_hist and args.dist == "fixed" and not args.lossy:
        roofline["valu_issue"] = read_valu_issue(bases, times, args.steps)
    pipe = {"algo_bytes": step_algo, "step_ms": round(step_s * 1e3, 3),
            "kernel_ms_sum": round(sum(kern[k]["ms_avg"] * kern[k]["launches"] / args.steps for k in kern if k != "k_synth"), 3),
            "GBps": round(step_algo / step_s / 1e9, 1),
            "encoder": "two pass (sizes, encode)" if args.twopass else
                       ("one pass (sizes from the entries' own histograms, records written in place)" if by_hist else
                        "one pass (scratch slots, compaction on a second stream)")}
    pipe["frac"] = round(pipe["GBps"] / HBM_PEAK_GBS, 4)

    cpu_res = None
    if cpu:
        cpu_res = cpu_baseline(ctx, api, d_text, off, lens, hlen, args, state,
                               dict(p_out=p_out, p_rec=p_rec, n=n, movie=movie))

    which = "BASELINE.json configs[3]" if world == 1 else f"BASELINE.json configs[4] slice: {world * n} entries over {world} GPUs"
    line = {
        "metric": "dexqv encode input GB/s (5 QV/tag stream bytes per base; .dexqv bit-exact vs reference)",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"dexqv 5-stream Huffman encode, {n} x {args.mean} .quiva per GPU "
                               f"({args.dist} lengths, {which}), HBM-resident",
                   "entries_per_gpu": n, "mean_len": args.mean, "lossy": bool(args.lossy),
                   "run_density": {"del": args.del_run_p, "sub": args.sub_run_p},
                   "input_bytes_per_gpu": 5 * bases, "text_image_bytes_per_gpu": text_bytes,
                   "output_bytes": all_out, "ratio": round(5.0 * all_bases / all_out, 3),
                   "sharding": "contiguous entry ranges of one file (host-side scan state + 12 KB histogram sum, gloo; no RCCL)"
                               if world > 1 else "single GPU"},
        "roofline": roofline,
        "cpu_baseline": cpu_res,
        "roundtrip_bit_exact": roundtrip,                         # (lossy: against the text with QV.c:1355-1372's rounding applied)
        "tables_identical_across_ranks": tables_same,
        "host_table_build_us": state.get("host_build_us"),
        "encoder_route": dict(state.get("route") or {}, scratch_budget_bytes=int(budget_gb * 1e9) or None,
                              device_bytes_in_use_after_step=state.get("mem_used_peak")),
        "decode": state.get("decode"),
        "decode_indexed": state.get("decode_indexed"),
        "decode_walk_indexed": state.get("decode_walk_indexed"),
        "device_walk": state.get("device_walk"),
        "text_front_end": fr,
        "pipeline": pipe,
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in kern.items()},
    }
    if world > 1:
        # what ONE GPU does with the same slice (2.5 M entries) when it runs alone: the figure a scaling efficiency is to be read
        # against -- the N = 1 line of this bench is BASELINE configs[3], 1 M entries, a different per-GPU workload (measured by
        # tools/evidence.sh under WORLD_SIZE=1 --entries 2500000, kept in profiles/per_gpu_reference.json)
        try:
            ref = json.load(open(os.path.join(ROOT, "profiles", "per_gpu_reference.json")))
            if ref.get("entries_per_gpu") == n and ref.get("mean_len") == args.mean:
                line["per_gpu_reference"] = ref
        except Exception:
            line["per_gpu_reference"] = None
        dist.destroy_process_group()
    del d_text, p_out, state["p_out"], p_text
    ctx.close()
    torch.cuda.empty_cache()
    return line


def read_traffic(path, workload, kernel, want, launches_per_step=1, per_step=False):
    """HBM bytes per launch (per_step: per bench step) of the kernels timed under bench id `kernel`, from the
    committed PMC summary (profiles/traffic.json, "bench_ids"), when it was collected on the same workload shape and
    with the same number of launches per step; None otherwise."""
    try:
        tf = json.load(open(path))["workloads"][workload]
        if any(tf.get(k) != v for k, v in want.items()):
            return None
        e = tf["bench_ids"].get(kernel)
        if e is None or e["launches_per_step"] != launches_per_step:
            return None                                   # not profiled, or profiled with another grouping
        return e["hbm_bytes_per_step"] if per_step else e["hbm_bytes_per_launch"]
    except Exception:
        return None


VALU_CYCLES = 4.15        # cycles a SIMD spends on a wave64 instruction of the kinds the kernels are made of (profiles/r05_valu_issue.txt)


def read_valu_issue(bases, times_ms, steps):
    """What the step's main kernels cost in vector instruction issue: SQ_INSTS_VALU of the committed evidence set's counter pass
    (profiles/<tag>_sq_counters.json: a batch of 200 k x 10 kb; the counts go with the symbols) x 4.15 cycles / (1024 SIMDs x 2.4 GHz),
    beside the kernels' measured times.  A constant of the tree, like roofline.traffic -- None when the set has no counters."""
    try:
        man = json.load(open(os.path.join(ROOT, "profiles", "MANIFEST.json")))
        sq = json.load(open(os.path.join(ROOT, "profiles", man["files"]["sq_counters.json"])))
        n0, m0 = (int(x) for x in sq["batch"].split(" x "))
        scale = bases / float(n0 * m0)
        out, tot_issue, tot_ms = {}, 0.0, 0.0
        for dev, bid in (("k_qv_hist", "k_qv_hist"), ("k_qv_sizes_hist", "k_qv_sizes"), ("k_qv_encode_fast", "k_qv_encode")):
            insts = sq["per_launch"][dev]["SQ_INSTS_VALU"] * scale
            ms_issue = insts * VALU_CYCLES / (1024 * 2.4e9) * 1e3
            ms = times_ms.get(bid, (0.0, 0))[0] / steps
            out[dev] = {"valu_insts_per_step": round(insts), "ms_of_issue": round(ms_issue, 2), "ms_measured": round(ms, 2),
                        "frac_of_kernel_time": round(ms_issue / ms, 3) if ms else None}
            tot_issue += ms_issue; tot_ms += ms
        return {"cycles_per_instruction_and_simd": VALU_CYCLES, "peak_wave64_insts_per_s": round(1024 * 2.4e9 / VALU_CYCLES),
                "per_kernel": out, "ms_of_issue_per_step": round(tot_issue, 2), "frac_of_the_kernels_time": round(tot_issue / tot_ms, 3) if tot_ms else None,
                "source": "profiles/" + man["files"]["sq_counters.json"] + " scaled by symbols; issue cost: profiles/r05_valu_issue.txt (tools/microbench/valu_issue.hip)",
                "note": "the step's kernels are held by vector instruction issue more than by the HBM: the roofline above stays SURVEY 8(d)'s"}
    except Exception:
        return None


def cpu_baseline(ctx, api, d_text, off, lens, hlen, args, state, big):
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
    for tool, srcname, payload, expect in (("dexqv", "s.quiva", sample, want), ("undexqv", "s.dexqv", want, None)):
        cli = os.path.join(ROOT, "dextractor_amd", "bin", tool)
        if not os.path.isfile(cli):
            continue
        with tempfile.TemporaryDirectory(dir=shm) as d:
            src = os.path.join(d, srcname)
            with open(src, "wb") as f:
                f.write(payload)
            flags = ["-k"] + (["-l"] if args.lossy and tool == "dexqv" else []) + (["-U"] if tool == "undexqv" else [])
            dtc, runs = None, []
            for _ in range(3):                     # best of three (the first start of a tool on a box pages its libraries in)
                t0 = time.perf_counter()
                rc_ = subprocess.call([cli] + flags + [src])
                runs.append(round(time.perf_counter() - t0, 3))
                dtc = runs[-1] if dtc is None else min(dtc, runs[-1])
                if rc_ != 0:
                    break
            same = False
            if rc_ == 0:
                with open(os.path.join(d, "s.dexqv" if tool == "dexqv" else "s.quiva"), "rb") as f:
                    back = f.read()
                if expect is not None:
                    same = back == expect
                else:                      # decoded text: the data lines (the synthetic headers are fixed-width, undexqv's are not)
                    data = lambda b_: [ln for ln in b_.split(b"\n") if not ln.startswith(b"@")]
                    same = args.lossy or data(back) == data(sample)
        res["cli_end_to_end" if tool == "dexqv" else "cli_undexqv_end_to_end"] = {
            "seconds": round(dtc, 3), "GBps": round(5 * sbases / dtc / 1e9, 3), "output_identical": bool(same),
            "runs_s": runs, "note": "process start to exit, tmpfs to tmpfs, best of the runs; this process holds the GPU meanwhile"}

    # ... and on a LARGE file (tools/cli_scale.py: a 20 GB .quiva and its 4 GB .fasta, made, packed, unpacked and compared in a
    # process of its own; per-stage marks of the tools in "marks_ms")
    if args.cli_large_gb > 0:
        trace("cpu_baseline: the tools on a large file")
        big_dir = tempfile.mkdtemp(prefix="cliscale.", dir=shm)      # ours to remove: a timeout's SIGKILL skips the script's own cleanup
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_scale.py"), str(args.cli_large_gb), "--dir", big_dir],
                               capture_output=True, timeout=900)
            big_cli = json.loads(r.stdout.decode().strip().splitlines()[-1])
            if "skipped" in big_cli:
                raise RuntimeError(big_cli["skipped"])
            brief = {"file": big_cli["file"]}
            for tool_, runs_ in big_cli["runs"].items():
                best = min(runs_, key=lambda x: x["s"])
                brief[tool_] = {k_: v_ for k_, v_ in best.items() if k_ != "marks_ms"}
                brief[tool_]["stages_ms"] = [m_ for m_ in best["marks_ms"] if m_[1] not in ("start", "set device", "device properties", "events")]
                brief[tool_]["all_runs_s"] = [x["s"] for x in runs_]
                ok_ = [x.get("round_trip_identical") for x in runs_ if x.get("round_trip_identical") is not None]
                brief[tool_]["round_trip_identical"] = bool(ok_) and all(ok_) if tool_.startswith("un") else None
            res["cli_end_to_end_large"] = brief
        except Exception as e:
            res["cli_end_to_end_large"] = {"skipped": f"{type(e).__name__}: {e}"[:300]}
        finally:
            shutil.rmtree(big_dir, ignore_errors=True)

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


# ------------------------------------------------------------------------------------------------
#  dexta / dexar (BASELINE configs[1] / [2])
# ------------------------------------------------------------------------------------------------
def pack2_bench(args, arrow):
    """2-bit pack + unpack of `--reads` x `--mean` reads (80-column text).  A host-generated tile of
    20000 reads is replicated on the device to the full size.  Returns the JSON line as a dict."""
    import torch
    from dextractor_amd import _lib as L
    from dextractor_amd import api, synth
    torch.cuda.set_device(0)
    ctx = api.Context(0)
    name = "dexar" if arrow else "dexta"
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

    up = lambda a, dt: Ptr(torch.from_numpy(np.ascontiguousarray(a).view(dt)).cuda())
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    p_text, p_off, p_tlen, p_nsym = Ptr(d_text), up(off, np.int64), up(tlen, np.int32), up(nsym, np.int32)
    p_hdr, p_hoff, p_ooff = up(blob, np.uint8), up(hoff, np.int64), up(ooff, np.int64)
    p_out = Ptr(torch.empty(out_bytes + 64, dtype=torch.uint8, device="cuda"))
    # decode side: packed bytes sit after each record's framing; text goes back in 80-column lines
    ioff = (ooff + (hoff[1:] - hoff[:-1])).astype(np.uint64)
    p_ioff = up(ioff, np.int64)
    p_boff = p_off
    if not args.p2_align:
        p_back = Ptr(torch.zeros(n * tb // n0 + 64, dtype=torch.uint8, device="cuda"))
    else:                                             # (experiment: what misaligned 16-byte stores cost the unpack kernel)
        A = np.uint64(args.p2_align)
        tl = tlen.astype(np.uint64)
        boff = np.concatenate([[0], np.cumsum((tl + A - np.uint64(1)) // A * A)[:-1]]).astype(np.uint64)
        p_back = Ptr(torch.zeros(int(boff[-1] + tl[-1]) + 4096, dtype=torch.uint8, device="cuda"))
        p_boff = up(boff, np.int64)
    alpha = L.DX_ALPHA_ARROW if arrow else L.DX_ALPHA_BASES
    letters = L.DX_LETTERS_ARROW if arrow else L.DX_LETTERS_UPPER

    def step():
        ctx.pack2_encode(alpha, p_text, p_off, p_tlen, p_nsym, n, p_hdr, p_hoff, p_out, p_ooff)
        ctx.pack2_decode(letters, p_out, p_ioff, p_nsym, n, 80, p_back, p_boff)

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
    for t in sorted({0, reps - 1} if not args.p2_align else set()):
        a = d_text[t * tb: (t + 1) * tb].cpu().numpy()
        b = p_back.t[t * tb: (t + 1) * tb].cpu().numpy()
        same = same and bool(np.array_equal(a[seqmask], b[seqmask]))
    bases = int(nsym.astype(np.uint64).sum())
    text_in = int(tlen.astype(np.uint64).sum())
    enc_ms, dec_ms = times["k_pack2_encode"][0] / args.steps, times["k_pack2_decode"][0] / args.steps
    algo_enc, algo_dec = text_in + out_bytes, (out_bytes + text_in)

    # CPU baseline: the real reference tool on the host tile (one core), and the GPU file driver on the
    # same tile compared with its output byte for byte
    cpu = None
    if not args.no_cpu_baseline:
        ref_bin = os.path.join(ROOT, "oracle", "_ref", name)
        shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
        ext = ".arrow" if arrow else ".fasta"
        cpu = {"unit": "GB/s", "cores": 1, "sample": f"{n0} reads of the bench tile ({tb / 1e9:.2f} GB of text)"}
        want = None
        if os.path.isfile(ref_bin):
            with tempfile.TemporaryDirectory(dir=shm) as d:
                src = os.path.join(d, "s" + ext)
                with open(src, "wb") as f:
                    f.write(tile.text)
                t1 = time.perf_counter()
                subprocess.check_call([ref_bin, "-k", src])
                dtc = time.perf_counter() - t1
                with open(os.path.join(d, "s." + name), "rb") as f:
                    want = f.read()
            cpu.update(kind="reference", value=round(tb / dtc / 1e9, 4), seconds=round(dtc, 2))
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle as O
            t1 = time.perf_counter()
            want = (O.dexar if arrow else O.dexta)(tile.text)
            dtc = time.perf_counter() - t1
            cpu.update(kind="port", value=round(tb / dtc / 1e9, 4), seconds=round(dtc, 2))
        got = (ctx.dexar if arrow else ctx.dexta)(tile.text)
        cpu["gpu_output_identical"] = bool(got == want)

    tr = read_traffic(args.traffic_file, name, "k_pack2_encode", dict(reads=n, mean=args.mean))
    line = {"metric": f"{name} 2-bit pack input GB/s (+ unpack); round-trip bit-exact",
            "value": round(text_in / (enc_ms * 1e-3) / 1e9, 2), "unit": "GB/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{name} + un{name}, {n} x {args.mean} reads, 80-column text, HBM-resident "
                                   f"(BASELINE.json configs[{2 if arrow else 1}])", "reads": n, "bases": bases,
                       "text_bytes": text_in, "packed_bytes": out_bytes},
            "roofline": {"kernel": "k_pack2_encode", "bound": "hbm", "achieved": round(algo_enc / (enc_ms * 1e-3) / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo_enc / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": tr, "traffic_source": "PMC passes of the committed evidence set (" + os.path.relpath(args.traffic_file, ROOT) + "): a constant of the tree, not a measurement of this run" if tr is not None else None,
                         "algo_bytes_per_launch": algo_enc},
            "decode": {"kernel": "k_pack2_decode", "ms": round(dec_ms, 3), "GBps": round(algo_dec / (dec_ms * 1e-3) / 1e9, 1),
                       "frac": round(algo_dec / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": read_traffic(args.traffic_file, name, "k_pack2_decode", dict(reads=n, mean=args.mean))},
            "encode_ms": round(enc_ms, 3), "roundtrip_bit_exact": same, "cpu_baseline": cpu}
    del d_text, p_text, p_out, p_back
    ctx.close()
    torch.cuda.empty_cache()
    return line


if __name__ == "__main__":
    main()
