#!/usr/bin/env python3
"""Detail pass of scripts/x6_bias_ab.py for the diagnostic library: WHICH outputs are wrong and what value the epilogue used as bias."""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch  # noqa: E402
from bem import native, ops  # noqa: E402

print("library:", native.LIB_PATH)
torch.manual_seed(0)
side = torch.cuda.Stream()
big = torch.randn(64, 320, 128, 128, device="cuda")
dww = torch.randn(320, 1, 3, 3, device="cuda")
for K, M in ((40, 320), (160, 40)):
    B, H, W = 4, 224, 320
    L = H * W
    x = torch.randn(B, K, H, W, device="cuda")
    w = torch.randn(M, K, device="cuda") * K ** -0.5
    b = torch.arange(M, device="cuda", dtype=torch.float32) + 100.0          # bias value identifies the row it belongs to
    nob = torch.einsum("mk,bkhw->bmhw", w.double(), x.double()).float()
    Wp = ops.pack_pw_weight(w, x6=True)
    seen = 0
    for rep in range(40):
        with torch.cuda.stream(side):
            ops.dwconv3x3(big, dww, None, mode=2)
        y = ops.pw_gemm(x, Wp, M, bias=b)
        used = (y - nob).reshape(B, M, L)                                     # the bias value each output actually received
        bad = (used - b[None, :, None]).abs() > 0.5
        idx = bad.nonzero()
        if idx.shape[0] and seen < 3:
            seen += 1
            bi, ri, pi = idx[:, 0].cpu(), idx[:, 1].cpu(), idx[:, 2].cpu()
            u = used[bad].cpu()
            print(f"K={K} M={M} rep {rep}: {idx.shape[0]} bad outputs")
            print("   row % 32 histogram:", sorted(Counter((ri % 32).tolist()).items()))
            print("   row // 32 histogram:", sorted(Counter((ri // 32).tolist()).items()))
            print("   pixel % 128 histogram (first 12):", sorted(Counter((pi % 128).tolist()).items())[:12], "...")
            print("   distinct (pixel // 128) tiles:", len(set((pi // 128).tolist())), "batches:", sorted(set(bi.tolist())))
            d = (u - b[ri.cuda()].cpu()).round()
            print("   (bias used - bias expected) histogram:", sorted(Counter(d.tolist()).items())[:16])
            for j in range(min(6, idx.shape[0])):
                print(f"      b={int(bi[j])} row={int(ri[j])} px={int(pi[j])} used={float(u[j]):.3f} expected={float(b[ri[j]]):.1f}")
    torch.cuda.synchronize()
