import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
from bem import ops
torch.manual_seed(0)
for K, M, H, W, use_res, use_b in ((40, 40, 224, 320, True, True), (40, 40, 224, 320, False, True), (160, 40, 224, 320, True, True), (40, 320, 224, 320, False, True), (40, 40, 128, 128, True, True)):
    x = torch.randn(1, K, H, W, device="cuda"); w = torch.randn(M, K, device="cuda") * K ** -0.5
    res = torch.randn(1, M, H, W, device="cuda") if use_res else None
    bias = torch.randn(M, device="cuda") if use_b else None
    ref = torch.einsum("mk,bkhw->bmhw", w, x) + (res if use_res else 0) + (bias[None, :, None, None] if use_b else 0)
    y = ops.pw_gemm(x, ops.pack_pw_weight(w, x6=True), M, res=res, bias=bias)
    bad = ((y - ref).abs() > 1e-3).reshape(M, H * W)
    idx = bad.nonzero()
    print(K, M, H, W, "res", use_res, "bias", use_b, "bad", idx.shape[0])
    if idx.shape[0]:
        rows = sorted(set(idx[:, 0].tolist())); ps = sorted(set(idx[:, 1].tolist()))
        print("  rows", rows[:20], "n", len(rows)); print("  px min/max", ps[0], ps[-1], "n", len(ps), "first", ps[:12])
        i0 = idx[0]; print("  sample y", y.reshape(M, -1)[i0[0], i0[1]].item(), "ref", ref.reshape(M, -1)[i0[0], i0[1]].item(), "res", res.reshape(M,-1)[i0[0], i0[1]].item() if use_res else None)
