"""Sporadic-corruption probe for the kernels of round 2 that feed LDS-returned register pairs to packed math (DESIGN.md section 6.4 family):
bem_pi_gate_x6_f32 is deterministic, so every repeat of a launch must reproduce the first one bit for bit; run under the two-workgroups-
per-CU load of the bench shape.  Also repeats bem_pw_wgrad_x6_f32 (its second-stage atomics make it order-dependent: compared at 1e-5)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-enhancement-model_amd"))
from bem import ops
g = torch.Generator().manual_seed(0)
d = lambda t: t.cuda()
bad_total = 0
for (B, C, Hd, H, W) in [(64, 40, 160, 128, 128), (16, 40, 160, 128, 128), (64, 24, 96, 64, 64)]:
    x = d(torch.randn(B, C, H, W, generator=g)); lw = d(1 + 0.1 * torch.randn(C, generator=g)); lb = d(0.1 * torch.randn(C, generator=g))
    wi = d(torch.randn(2 * Hd, C, generator=g) * C ** -0.5); bi = d(torch.randn(2 * Hd, generator=g))
    wd = d(torch.randn(2 * Hd, 1, 3, 3, generator=g) / 3); bd = d(torch.randn(2 * Hd, generator=g))
    perm = ops.gate_order(Hd, "cuda")
    Wg = ops.pack_pw_weight(wi[perm].contiguous(), x6=True); bg = bi[perm].contiguous()
    wdg, bdg = ops.dw_gate_params(wd, bd, Hd)
    ref = ops.pi_gate(x, lw, lb, 1e-6, Wg, bg, wdg, bdg, Hd).clone()
    bad = 0
    for it in range(150):
        y = ops.pi_gate(x, lw, lb, 1e-6, Wg, bg, wdg, bdg, Hd)
        bad += int((y != ref).sum())
    print(f"pi_gate B={B} C={C} {H}x{W}: {bad} differing outputs in 150 launches of {ref.numel()} outputs")
    bad_total += bad
for (B, M, K, L) in [(16, 320, 40, 16384), (16, 40, 160, 16384), (16, 80, 80, 4096)]:
    dy = d(torch.randn(B, M, L, generator=g)); xx = d(torch.randn(B, K, L, generator=g))
    dw0 = torch.zeros(M, K, device="cuda"); ops.pw_wgrad_(dy, xx, dw0)
    scale = float(dw0.abs().max()); bad = 0
    for it in range(100):
        dw = torch.zeros(M, K, device="cuda"); ops.pw_wgrad_(dy, xx, dw)
        bad += int(((dw - dw0).abs() > 1e-5 * scale).sum())
    print(f"wgrad_x6 M={M} K={K} L={L}: {bad} outputs off by more than 1e-5 of the largest in 100 launches")
    bad_total += bad
print("TOTAL BAD", bad_total)
sys.exit(1 if bad_total else 0)
