"""Micro-benchmark of the depthwise 3x3 kernel at the Stage-II shapes (gate and SiLU forms, per-sample weights)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for C, H, mode in ((320, 128, 2), (40, 128, 1), (640, 64, 2), (80, 64, 1), (1280, 32, 2)):
    x = torch.randn(B, C, H, H, device="cuda")
    w = torch.randn(B, C, 1, 3, 3, device="cuda") / 3
    b = torch.randn(B, C, device="cuda")
    f = lambda: ops.dwconv3x3(x, w, b, mode)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    by = 4.0 * x.numel() * (1.5 if mode == 2 else 2.0)
    print(f"C={C} H={H} mode={mode}: {dt*1e6:.0f} us  {by/dt/1e12:.2f} TB/s algorithmic")
