"""Run-to-run determinism of the x6 streaming GEMM at the decomposition's attention-fuse shape (debugging aid)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
from bem import ops
g = torch.Generator().manual_seed(0)
for (L, M, K1, K2, mode) in ((71680, 32, 32, 32, 2), (71680, 32, 64, 0, 0), (71680, 40, 32, 32, 2), (57344, 32, 32, 32, 2), (131072, 32, 32, 32, 2), (71680, 32, 16, 16, 2)):
    x1 = torch.randn(1, K1, L, generator=g).cuda()
    x2 = torch.randn(1, K2, L, generator=g).cuda() if K2 else None
    W = torch.randn(M, K1 + K2, generator=g).cuda() * 0.1
    bias = torch.randn(M, generator=g).cuda()
    Wp = ops.pack_pw_weight(W, x6=True)
    ref = (W.double() @ torch.cat([x1, x2], 1)[0].double() if K2 else W.double() @ x1[0].double()) + bias.double()[:, None]
    outs = [ops.pw_gemm(x1, Wp, M, x2=x2, in_mode=mode, bias=bias) for _ in range(6)]
    torch.cuda.synchronize()
    bad = [(o[0].double() - ref).abs() > 1e-3 for o in outs]
    nb = [int(b.sum()) for b in bad]
    msg = f"L={L} M={M} K={K1}+{K2} mode={mode}: wrong elements per run {nb}"
    if any(nb):
        b = bad[nb.index(max(nb))]
        rows, px = b.nonzero(as_tuple=True)
        msg += f"  rows {sorted(set(rows.tolist()))[:12]}  pixels mod 256: {sorted(set((px % 256).tolist()))[:20]}  tiles {sorted(set((px // 256).tolist()))[:12]}"
        o = outs[nb.index(max(nb))][0]
        i = (rows[0].item(), px[0].item())
        msg += f"  e.g. got {o[i].item():.4f} want {ref[i].item():.4f} bias {bias[i[0]].item():.4f}"
    print(msg, flush=True)
