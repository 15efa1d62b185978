"""Micro-benchmark of the MFMA convolutions at the Stage-II shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for Ci, Co, H, k, st in ((32, 32, 128, 3, 1), (32, 40, 128, 3, 1), (40, 16, 128, 3, 1), (40, 80, 128, 4, 2), (80, 160, 64, 4, 2)):
    x = torch.randn(B, Ci, H, H, device="cuda")
    w = torch.randn(Co, Ci, k, k, device="cuda") / (Ci * k * k) ** 0.5
    b = torch.randn(Co, device="cuda")
    f = lambda: ops.conv2d(x, w, b, stride=st, pad=1)
    y = f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    by = 4.0 * (x.numel() + y.numel())
    fl = 2.0 * y.numel() * Ci * k * k
    print(f"Cin={Ci} Cout={Co} H={H} k={k} s={st}: {dt*1e6:.0f} us  {by/dt/1e12:.2f} TB/s  {fl/dt/1e12:.1f} TF/s")
