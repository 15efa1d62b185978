import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 16
for C, H, W, mode in ((320, 224, 320, 2), (40, 224, 320, 1), (640, 112, 160, 2)):
    x = torch.randn(B, C, H, W, device="cuda"); w = torch.randn(B, C, 1, 3, 3, device="cuda") / 3; b = torch.randn(B, C, device="cuda")
    f = lambda: ops.dwconv3x3(x, w, b, mode)
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    by = 4.0 * x.numel() * (1.5 if mode == 2 else 2.0)
    print(f"C={C} {H}x{W} mode={mode}: {dt*1e6:.0f} us  {by/dt/1e12:.2f} TB/s")
