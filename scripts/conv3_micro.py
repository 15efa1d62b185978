"""The 3x3 convs of the eval step at the bench shapes: row-form x6 kernel (conv_rows_x6.hip) vs the nine shifted taps (BEM_CONV3_ROWS=0).
   python scripts/conv3_micro.py [reps]"""
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


print("BEM_CONV3_ROWS =", os.environ.get("BEM_CONV3_ROWS", "1"))
for (B, Ci, Co, H, W) in [(64, 32, 32, 128, 128), (64, 32, 40, 128, 128), (64, 40, 16, 128, 128), (8, 32, 32, 128, 128), (16, 32, 32, 128, 128)]:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g).cuda(); w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).cuda(); b = torch.randn(Co, generator=g).cuda()
    t = timeit(lambda: ops.conv2d(x, w, b, stride=1, pad=1))
    fl = 2.0 * B * Co * Ci * 9 * H * W
    print(f"B={B} {Ci}->{Co} {H}x{W}: {t:8.1f} us   {fl / t / 1e6:6.1f} TFLOP/s (f32 conv flops)   {4.0 * B * (Ci + Co) * H * W / t / 1e3:7.1f} GB/s algorithmic")
