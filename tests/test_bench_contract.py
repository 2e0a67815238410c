"""bench.py's contract with the driver: one JSON line with the agreed fields, at N = 1 and -- every rank on device 0
of a one-GPU box -- through the self-launched N = 2 path (gloo exchange of the scan state and the histograms)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
          "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _run(extra, env=None):
    e = dict(os.environ, **(env or {}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, cwd=ROOT, timeout=600,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                         # exactly one JSON line on stdout
    return json.loads(lines[0])


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, timeout=120,
                         env=dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode != 0 and b"--gpus 2 but WORLD_SIZE=4" in out.stderr


@pytest.mark.gpu
def test_bench_line_single_gpu():
    d = _run(["--entries", "30000", "--steps", "2", "--warmup", "1", "--only-main", "--no-cpu-baseline"])
    assert all(k in d for k in FIELDS)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GB/s" and d["dtype"] == "u8"
    assert d["roundtrip_bit_exact"] is True and d["decode_indexed"]["bit_exact"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1
    # the whole step -- (4 + 5) bytes per base read + the output written -- over the step's wall time
    bases, out = 30000 * 10000, d["config"]["output_bytes"]
    assert abs(r["algo_bytes_per_step"] - (9 * bases + out)) < 1 and abs(r["achieved"] - (9 * bases + out) / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * r["achieved"]
    assert r["dominant_kernel"]["kernel"] in ("k_qv_hist", "k_qv_encode") and 0 < r["dominant_kernel"]["frac"] < 1
    assert "k_qv_compact" not in r["per_kernel"]                                    # no scratch slots, no compaction: the sizes come
    assert d["encoder_route"]["direct"] == 2 and d["encoder_route"]["groups"] == 0  # from the entries' own histograms (k_qv_sizes_hist)
    assert r["per_kernel"]["k_qv_sizes"]["algo_bytes_8d_per_step"] == 0
    assert abs(d["value"] - 5 * 30000 * 10000 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]


@pytest.mark.gpu
def test_bench_line_two_ranks_on_one_device():
    d = _run(["--gpus", "2", "--entries", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"DEXGPU_BENCH_ONE_DEVICE": "1"})
    assert all(k in d for k in FIELDS)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["tables_identical_across_ranks"] is True and d["roundtrip_bit_exact"] is True
    assert abs(d["value"] - 2 * 5 * 20000 * 10000 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]     # whole job


@pytest.mark.gpu
def test_bench_line_four_ranks_on_one_device():
    """The N = 4 launch of the driver's scaling run, rehearsed with every rank on device 0 (a GPU box admits six processes on
    its card at once, so the N = 8 launch cannot be rehearsed there: its sharding logic runs under gloo in
    tests/test_shard_gloo.py::test_eight_rank_sharded_dexqv_equals_single): one JSON line, n_gpus 4, identical tables on
    every rank, value = the whole job's bytes over the slowest rank's time."""
    d = _run(["--gpus", "4", "--entries", "8000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
             env={"DEXGPU_BENCH_ONE_DEVICE": "1"})
    assert all(k in d for k in FIELDS)
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["sharding"] != "single GPU"
    assert d["tables_identical_across_ranks"] is True and d["roundtrip_bit_exact"] is True
    assert abs(d["value"] - 4 * 5 * 8000 * 10000 / (d["ms_per_step"] * 1e-3) / 1e9) < 0.02 * d["value"]     # whole job
