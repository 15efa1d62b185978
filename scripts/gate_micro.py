import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B = 64
for C, Hd, H in ((40, 160, 128), (80, 320, 64), (160, 640, 32)):
    h = ops.empty_padded((B, 2 * Hd, H, H), "cuda"); h.normal_()
    wd = torch.randn(B, 2 * Hd, 1, 3, 3, device="cuda") / 3; bd = torch.randn(B, 2 * Hd, device="cuda")
    Wp = ops.pack_pw_weight(torch.randn(B, C, Hd, device="cuda") * Hd ** -0.5, x6=True); bo = torch.randn(B, C, device="cuda")
    res = torch.randn(B, C, H, H, device="cuda")
    def chain():
        return ops.pw_gemm(ops.dwconv3x3(h, wd, bd, 2), Wp, C, bias=bo, res=res)
    def fused():
        return ops.gate_proj(h, wd, bd, Wp, C, bias=bo, res=res)
    for name, f in (("chain", chain), ("fused", fused)):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"C={C} Hd={Hd} H={H} {name}: {dt*1e6:.0f} us")
