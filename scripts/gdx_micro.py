"""bem_gdmlp_x6_f32 at the bench's level-0 / level-1 shapes: parity against the unfused chain, time of the fused kernel, of the chain's kernels
(BEM_GDX_WPS=2: the C <= 48 form at two workgroups per CU).   python scripts/gdx_micro.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
d = lambda t: t.cuda()


def timeit(fn, n=reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, C, H, W) in [(64, 40, 128, 128), (64, 80, 64, 64), (64, 160, 32, 32)]:
    if C > 80: continue
    Hd = 4 * C
    g = torch.Generator().manual_seed(0)
    x = d(torch.randn(B, C, H, W, generator=g)); lw = d(1 + 0.1 * torch.randn(C, generator=g)); lb = d(0.1 * torch.randn(C, generator=g))
    wi = d(torch.randn(2 * Hd, C, generator=g) * C ** -0.5); bi = d(torch.randn(2 * Hd, generator=g))
    wd = d(torch.randn(2 * Hd, 1, 3, 3, generator=g) / 3); bd = d(torch.randn(2 * Hd, generator=g))
    wo = d(torch.randn(C, Hd, generator=g) * Hd ** -0.5); bo = d(torch.randn(C, generator=g))
    perm = ops.gate_interleave(Hd, "cuda")
    Wg = ops.pack_pw_weight(wi[perm].contiguous(), x6=True); bg = bi[perm].contiguous()
    Wo = ops.pack_pw_weight(wo, x6=True); Wi = ops.pack_pw_weight(wi, x6=True)
    w10 = ops.dw_gate_params10(wd, bd, Hd)
    fused = lambda: ops.gdmlp_x6(x, lw, lb, 1e-6, Wg, bg, w10, Wo, bo, Hd)
    def chain():
        t = ops.pw_gemm(x, Wi, 2 * Hd, ln=(lw, lb), ln_eps=1e-6, bias=bi)
        return ops.pw_gemm(ops.dwconv3x3(t, wd, bd, 2), Wo, C, bias=bo, res=x)
    y, r = fused(), chain()
    err = (y - r).abs().max().item()
    print(f"C={C} {H}x{W}: max |fused - chain| = {err:.3e} (|chain| max {r.abs().max().item():.2f})")
    print(f"  chain (3 kernels): {timeit(chain):8.1f} us")
    print(f"  fused: {timeit(fused):8.1f} us")
