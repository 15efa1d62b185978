"""Micro-benchmark of the fused SS2D scan at the Stage-II level shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for C, H in ((40, 128), (80, 64), (160, 32)):
    L, R = H * H, (C + 15) // 16
    x0, x1 = torch.randn(B, C, L, device="cuda"), torch.randn(B, C, L, device="cuda")
    xd0, xd1 = torch.randn(B, 2, R + 2, L, device="cuda"), torch.randn(B, 2, R + 2, L, device="cuda")
    dtw, dtb = torch.randn(4, C, R, device="cuda") * 0.3, torch.randn(4, C, device="cuda")
    A, Ds = -torch.rand(4 * C, device="cuda"), torch.ones(4 * C, device="cuda")
    f = lambda: ops.ss2d_scan(x0, x1, xd0, xd1, dtw, dtb, A, Ds)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    by = 4.0 * (4 * x0.numel() + 2 * xd0.numel())
    print(f"C={C} L={L}: {dt*1e6:.0f} us  {by/dt/1e12:.2f} TB/s algorithmic")
