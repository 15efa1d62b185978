"""Micro-benchmark of the pointwise GEMM at the Stage-II level-0 shapes (timing experiments)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
B, H = 64, 128
for K, M, ln in ((40, 40, True), (40, 320, True), (40, 10, False), (160, 40, False)):
    x = torch.randn(B, K, H, H, device="cuda")
    Wp = ops.pack_pw_weight(torch.randn(M, K, device="cuda") * K ** -0.5)
    lnp = (torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")) if ln else None
    f = lambda: ops.pw_gemm(x, Wp, M, ln=lnp)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    by = 4.0 * B * H * H * (K + M)
    print(f"K={K} M={M} ln={ln}: {dt*1e6:.0f} us  {2.0*B*H*H*K*M/dt/1e12:.1f} TF/s  {by/dt/1e12:.2f} TB/s  dbg={os.environ.get('BEM_PW_DBG','0')}")
