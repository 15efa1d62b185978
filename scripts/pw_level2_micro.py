"""The 1x1 GEMMs of a level-2 VSSBlock (C = 160, 32x32 planes, 64 rows) and of level 1 (C = 80, 64x64) as the eval step calls them:
in_proj (LayerNorm prologue), x_proj, out_proj (LayerNorm of y0 + y1, residual).   BEM_X6_RES10=0: streaming two-sweep LayerNorm form at K = 160.
   python scripts/pw_level2_micro.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn, n=reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("BEM_X6_RES10 =", os.environ.get("BEM_X6_RES10", "1"))
B = 64
for C, H, R in ((160, 32, 10), (80, 64, 5), (40, 128, 3)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, H, generator=g).cuda(); y1 = torch.randn(B, C, H, H, generator=g).cuda()
    lw, lb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    Wi = ops.pack_pw_weight((torch.randn(C, C, generator=g) * C ** -0.5).cuda())
    Wx = ops.pack_pw_weight((torch.randn(4 * (R + 2), C, generator=g) * C ** -0.5).cuda())
    t_in = timeit(lambda: ops.pw_gemm(x, Wi, C, ln=(lw, lb), ln_eps=1e-6))
    t_x = timeit(lambda: ops.pw_gemm(x, Wx, 4 * (R + 2)))
    t_out = timeit(lambda: ops.pw_gemm(x, Wi, C, x2=y1, in_mode=1, ln=(lw, lb), ln_eps=1e-6, res=x))
    by = 4.0 * B * C * H * H
    print(f"C={C} {H}x{H}: in_proj {t_in:7.1f} us ({2 * by / t_in / 1e6:5.2f} TB/s)   x_proj {t_x:7.1f} us   out_proj {t_out:7.1f} us ({4 * by / t_out / 1e6:5.2f} TB/s)")
