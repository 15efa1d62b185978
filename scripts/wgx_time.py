"""time bem_pw_wgrad_x6_f32 (BEM_WGRAD_X6=0: the f32-MFMA form) at the training bench's level-0 shapes"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
g = torch.Generator().manual_seed(0)
for (B, M, K, L) in [(16, 320, 40, 16384), (16, 40, 40, 16384), (16, 40, 160, 16384)]:
    dy = torch.randn(B, M, L, generator=g).cuda(); x = torch.randn(B, K, L, generator=g).cuda()
    dw = torch.zeros(M, K, device="cuda"); db = torch.zeros(M, device="cuda")
    for dbg in [0, 0]:
        for _ in range(3): ops.pw_wgrad_(dy, x, dw, dbias=db)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.pw_wgrad_(dy, x, dw, dbias=db)
        e1.record(); torch.cuda.synchronize()
        print(f"M {M} K {K}: {e0.elapsed_time(e1) * 50:.1f} us   ({4 * (M + K) * B * L / 1e6:.0f} MB)")
