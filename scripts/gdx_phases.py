"""Timing anatomy of bem_gdmlp_x6_f32 (level-0 shape, C = 40, 128 x 128 planes): batch sizes that put 1 / 2 / 3 workgroups on a CU
(B = 2, 4, 6 -> 256, 512, 768 workgroups) and the bench's B = 64.   python scripts/gdx_phases.py [reps]"""
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


shapes = [(40, 128, 128), (80, 64, 64)]
for (C, H, W) in shapes:
    Hd = 4 * C
    g = torch.Generator().manual_seed(0)
    lw = d(1 + 0.1 * torch.randn(C, generator=g)); lb = d(0.1 * torch.randn(C, generator=g))
    wi = d(torch.randn(2 * Hd, C, generator=g) * C ** -0.5); bi = d(torch.randn(2 * Hd, generator=g))
    wd = d(torch.randn(2 * Hd, 1, 3, 3, generator=g) / 3); bd = d(torch.randn(2 * Hd, generator=g))
    wo = d(torch.randn(C, Hd, generator=g) * Hd ** -0.5); bo = d(torch.randn(C, generator=g))
    perm = ops.gate_interleave(Hd, "cuda")
    Wg = ops.pack_pw_weight(wi[perm].contiguous(), x6=True); bg = bi[perm].contiguous()
    w10 = ops.dw_gate_params10(wd, bd, Hd); Wo = ops.pack_pw_weight(wo, x6=True)
    tiles = (H // 4) * (W // 32)
    for B in sorted({max(1, 256 // tiles), max(1, 512 // tiles), max(1, 768 // tiles), 64}):
        x = d(torch.randn(B, C, H, W, generator=g))
        row = []
        for dbg in (0,):
            row.append(f"{timeit(lambda: ops.gdmlp_x6(x, lw, lb, 1e-6, Wg, bg, w10, Wo, bo, Hd)):7.1f}")
        print(f"C={C} {H}x{W} B={B:3d} ({B * tiles:5d} workgroups)  us  " + "  ".join(row), flush=True)
