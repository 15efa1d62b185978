"""Elementary accuracy of the two GEMM operand formats against a float64 reference."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
torch.manual_seed(0)
for K, M in ((40, 40), (160, 40), (40, 320), (320, 80)):
    x = torch.randn(2, K, 64, 64, device="cuda")
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    ref = torch.einsum("mk,bkhw->bmhw", w.double(), x.double())
    t32 = torch.einsum("mk,bkhw->bmhw", w, x)
    for name, x6 in (("f32-mfma", False), ("x6", True)):
        y = ops.pw_gemm(x, ops.pack_pw_weight(w, x6=x6), M)
        e = (y.double() - ref).abs()
        print(f"K={K} M={M} {name:9s}: max {e.max().item():.3e} mean {e.mean().item():.3e} bias {(y.double()-ref).mean().item():+.2e}")
    e = (t32.double() - ref).abs()
    print(f"K={K} M={M} torch-f32 : max {e.max().item():.3e} mean {e.mean().item():.3e}")
