"""The row-major SS2D scan (the two single-orientation launches of ss2d_scan_rows_kernel) at the Stage-II shapes.   python scripts/scan_rm_micro.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn, n=reps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


B = 64
for C, H in ((40, 128), (80, 64)):
    L, R = H * H, (C + 15) // 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, H, generator=g).cuda()
    xd0, xd1 = torch.randn(B, 2, R + 2, L, generator=g).cuda(), torch.randn(B, 2, R + 2, L, generator=g).cuda()
    dtw, dtb = (torch.randn(4, C, R, generator=g) * 0.3).cuda(), (torch.randn(4, C, generator=g) - 2).cuda()
    A, Ds = (-torch.rand(4 * C, generator=g)).cuda(), torch.ones(4 * C).cuda()
    t = timeit(lambda: ops.ss2d_scan_rm(x, xd0, xd1, dtw, dtb, A, Ds))
    by = 4.0 * (2 * x.numel() + 2 * xd0.numel() + 2 * x.numel())
    print(f"C={C} L={L}: {t:7.1f} us for both orientations   {by / t / 1e6:5.2f} TB/s algorithmic")
