"""Micro-benchmark of the streaming x6 GEMM at the level-1 / level-2 shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for K, M, H in ((320, 80, 64), (640, 160, 32), (160, 40, 128), (160, 80, 64)):
    x = torch.randn(B, K, H, H, device="cuda"); r = torch.randn(B, M, H, H, device="cuda")
    Wp = ops.pack_pw_weight(torch.randn(M, K, device="cuda") * K ** -0.5)
    f = lambda: ops.pw_gemm(x, Wp, M, res=r)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    by = 4.0 * B * H * H * (K + 2 * M)
    print(f"K={K} M={M} H={H}: {dt*1e6:.0f} us  {by/dt/1e12:.2f} TB/s algorithmic")
