#!/usr/bin/env python3
"""Element-wise stress of the LayerNorm-prologue x6 GEMMs (LN parameters come from LDS and feed packed-f32 VALU ops: the
instruction combination of DESIGN.md section 6.4) at the config-5 level-0/1/2 planes with two workgroups per CU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from bem import native, ops  # noqa: E402

print("library:", native.LIB_PATH)
torch.manual_seed(0)
side = torch.cuda.Stream()
big = torch.randn(64, 320, 128, 128, device="cuda")
dww = torch.randn(320, 1, 3, 3, device="cuda")
tot = 0
for K, M, H, W, two in ((40, 320, 224, 320, False), (40, 40, 224, 320, False), (40, 40, 224, 320, True), (80, 640, 112, 160, False), (80, 80, 112, 160, True),
                        (160, 1280, 56, 80, False), (160, 160, 56, 80, True)):
    B = 4
    x = torch.randn(B, K, H, W, device="cuda") * 1.5 + 0.3
    x2 = torch.randn(B, K, H, W, device="cuda") if two else None
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    b = torch.randn(M, device="cuda")
    g, be = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda") * 0.2
    xs = (x + x2) if two else x
    n = F.layer_norm(xs.permute(0, 2, 3, 1).double(), (K,), g.double(), be.double(), 1e-5).permute(0, 3, 1, 2)
    ref = (torch.einsum("mk,bkhw->bmhw", w.double(), n) + b[None, :, None, None]).float()
    Wp = ops.pack_pw_weight(w, x6=True)
    bad = 0
    for rep in range(20):
        with torch.cuda.stream(side):
            ops.dwconv3x3(big, dww, None, mode=2)
        y = ops.pw_gemm(x, Wp, M, x2=x2, in_mode=1 if two else 0, ln=(g, be), bias=b)
        bad += int(((y - ref).abs() > 2e-3).sum())
    tot += bad
    print(f"K={K} M={M} {H}x{W} sum={two}: {bad} outputs off by more than 2e-3 in 20 launches of {y.numel()}")
print("TOTAL BAD", tot)
