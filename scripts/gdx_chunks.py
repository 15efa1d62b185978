"""Fixed vs per-chunk cost of bem_gdmlp_x6_f32: the same pixel set with hidden widths Hd = 16 .. 320 (1 .. 20 chunks of 16 gate channels).
   python scripts/gdx_chunks.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = lambda t: t.cuda()


def timeit(fn, n=reps):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for (B, C, H, W) in [(64, 40, 128, 128), (64, 80, 64, 64), (2, 40, 128, 128)]:
    g = torch.Generator().manual_seed(0)
    NB = max(1, min(8, int(1.2e9 // (B * C * H * W * 4))))        # rotate over > 1 GB of inputs: no launch finds its x in the 256 MB infinity cache
    xs = [d(torch.randn(B, C, H, W, generator=g)) for _ in range(NB)]
    x = xs[0]; lw = d(1 + 0.1 * torch.randn(C, generator=g)); lb = d(0.1 * torch.randn(C, generator=g))
    row = []
    for Hd in (16, 32, 64, 160, 320):
        wi = d(torch.randn(2 * Hd, C, generator=g) * C ** -0.5); bi = d(torch.randn(2 * Hd, generator=g))
        wd = d(torch.randn(2 * Hd, 1, 3, 3, generator=g) / 3); bd = d(torch.randn(2 * Hd, generator=g))
        wo = d(torch.randn(C, Hd, generator=g) * Hd ** -0.5); bo = d(torch.randn(C, generator=g))
        perm = ops.gate_interleave(Hd, "cuda")
        Wg = ops.pack_pw_weight(wi[perm].contiguous(), x6=True); bg = bi[perm].contiguous()
        w10 = ops.dw_gate_params10(wd, bd, Hd); Wo = ops.pack_pw_weight(wo, x6=True)
        it = [0]
        def cold():
            it[0] += 1
            return ops.gdmlp_x6(xs[it[0] % NB], lw, lb, 1e-6, Wg, bg, w10, Wo, bo, Hd)
        row.append(f"Hd={Hd}: {timeit(lambda: ops.gdmlp_x6(x, lw, lb, 1e-6, Wg, bg, w10, Wo, bo, Hd)):7.1f} / cold {timeit(cold):7.1f}")
    print(f"B={B} C={C} {H}x{W}  us  " + "   ".join(row), flush=True)
