"""bem_ss2d_front_x6_f32 at the bench's level-0 shape against the three kernels it replaces.   python scripts/ss2d_front_micro.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = lambda t: t.cuda()


def timeit(fn, n=reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, C, R, H, W) in [(64, 40, 3, 128, 128), (16, 40, 3, 128, 128)]:
    Mx = 4 * (R + 2)
    g = torch.Generator().manual_seed(0)
    x = d(torch.randn(B, C, H, W, generator=g)); lw = d(1 + 0.1 * torch.randn(C, generator=g)); lb = d(0.1 * torch.randn(C, generator=g))
    Wpi = ops.pack_pw_weight(d(torch.randn(C, C, generator=g) * C ** -0.5), x6=True)
    wd = d(torch.randn(C, 1, 3, 3, generator=g) / 3)
    Wpx = ops.pack_pw_weight(d(torch.randn(Mx, C, generator=g) * C ** -0.5), x6=True)
    fused = lambda: ops.ss2d_front(x, lw, lb, 1e-6, Wpi, None, wd, None, Wpx, Mx)
    def chain():
        t = ops.pw_gemm(x, Wpi, C, ln=(lw, lb), ln_eps=1e-6)
        xc = ops.dwconv3x3(t, wd, None, 1)
        return xc, ops.pw_gemm(xc, Wpx, Mx)
    (a, b), (c, e) = fused(), chain()
    print(f"B={B} C={C} {H}x{W}: max |xc diff| {(a - c).abs().max().item():.2e}  max |xd diff| {(b - e).abs().max().item():.2e}   "
          f"chain {timeit(chain):7.1f} us   fused {timeit(fused):7.1f} us")
