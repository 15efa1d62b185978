"""time bem_pi_gate_x6_f32 at the bench's level-0 shape: python scripts/pig_time.py  (BEM_PIG_DBG selects diagnostic variants)"""
import os, sys, importlib, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
B, C, Hd, H, W = 64, 40, 160, 128, 128
g = torch.Generator().manual_seed(0)
d = lambda t: t.cuda()
x = d(torch.randn(B, C, H, W, generator=g)); lw = d(torch.ones(C)); lb = d(torch.zeros(C))
wi = d(torch.randn(2 * Hd, C, generator=g) * C ** -0.5); bi = d(torch.randn(2 * Hd, generator=g))
wd = d(torch.randn(2 * Hd, 1, 3, 3, generator=g) / 3); bd = d(torch.randn(2 * Hd, generator=g))
perm = ops.gate_order(Hd, "cuda")
Wg = ops.pack_pw_weight(wi[perm].contiguous(), x6=True); bg = bi[perm].contiguous()
wdg, bdg = ops.dw_gate_params(wd, bd, Hd)
for dbg in [0]:
    os.environ["BEM_PIG_DBG"] = str(dbg)
    for _ in range(3): ops.pi_gate(x, lw, lb, 1e-6, Wg, bg, wdg, bdg, Hd)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.pi_gate(x, lw, lb, 1e-6, Wg, bg, wdg, bdg, Hd)
    e1.record(); torch.cuda.synchronize()
    print(f"dbg {dbg} (1: no phase A, 2: no phase B, 4: no stores): {e0.elapsed_time(e1) * 100:.1f} us")
