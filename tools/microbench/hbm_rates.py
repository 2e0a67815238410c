"""HBM rates of plain torch kernels on this GPU, for reference beside the codec kernels (GPU box):
    python tools/microbench/hbm_rates.py
fill (write only), copy (read + write), sum (read only) over 16 GiB."""
import torch
n = 16 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: a.fill_(7));              print(f"fill  (write only)   {n / t / 1e12:.2f} TB/s")
t = timed(lambda: b.copy_(a));              print(f"copy  (read + write) {2 * n / t / 1e12:.2f} TB/s")
v = a.view(torch.int64)
t = timed(lambda: v.sum());                 print(f"sum   (read only)    {n / t / 1e12:.2f} TB/s")
