"""Micro-benchmark of the fused gdMlp kernel at the Stage-II level sizes (timing experiments)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for C, H in ((40, 128), (80, 64), (160, 32)):
    Hd = 4 * C
    x = torch.randn(B, C, H, H, device="cuda")
    Wpi = ops.pack_pw_weight_gate(torch.randn(2 * Hd, C, device="cuda") * C ** -0.5, Hd)
    Wpo = ops.pack_pw_weight(torch.randn(C, Hd, device="cuda") * Hd ** -0.5)
    bpi, dww, dwb, bpo = torch.randn(2 * Hd, device="cuda"), torch.randn(2 * Hd, 9, device="cuda"), torch.randn(2 * Hd, device="cuda"), torch.randn(C, device="cuda")
    lw, lb = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    f = lambda: ops.gdmlp_fused(x, lw, lb, 1e-5, Wpi, bpi, dww, dwb, Wpo, bpo, Hd)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    fl = 2.0 * B * H * H * (2 * Hd * C + Hd * C)
    print(f"C={C} H={H}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s useful  dbg={os.environ.get('BEM_GD_DBG','0')}")
