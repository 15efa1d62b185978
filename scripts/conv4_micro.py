"""The two 4x4 stride-2 down-sampling convs of the Stage-II net at the bench shapes: coalesced-row x6 kernel (conv4_x6.hip) against the
f32-MFMA im2col kernel it replaces (BEM_CONV4_FAST=0 in a second process).   python scripts/conv4_micro.py [reps]"""
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


print("BEM_CONV4_FAST =", ops.CONV4_FAST)
for (B, Ci, Co, H, W) in [(64, 40, 80, 128, 128), (64, 80, 160, 64, 64), (16, 40, 80, 128, 128), (16, 80, 160, 64, 64)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g).cuda(); w = (torch.randn(Co, Ci, 4, 4, generator=g) * (Ci * 16) ** -0.5).cuda(); b = torch.randn(Co, generator=g).cuda()
    t = timeit(lambda: ops.conv2d(x, w, b, stride=2, pad=1))
    fl = 2.0 * B * Co * Ci * 16 * (H // 2) * (W // 2)
    print(f"B={B} {Ci}->{Co} {H}x{W}: {t:8.1f} us   {fl / t / 1e6:6.1f} TFLOP/s (f32 conv flops)   {4.0 * B * (Ci * H * W + Co * H * W / 4) / t / 1e3:7.1f} GB/s algorithmic")
