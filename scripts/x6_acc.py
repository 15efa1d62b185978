"""Elementary accuracy of the two GEMM operand formats against a float64 reference, every kernel variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
import torch
import torch.nn.functional as F
from bem import ops
torch.manual_seed(0)
dev = "cuda"
def ref64(x, w, ln=None, bias=None, res=None):
    x = x.double()
    if ln is not None:
        x = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), ln[0].double(), ln[1].double(), 1e-5).permute(0, 3, 1, 2)
    y = torch.einsum("mk,bkhw->bmhw", w.double(), x)
    if bias is not None: y = y + bias.double()[None, :, None, None]
    if res is not None: y = y + res.double()
    return y
def ref32(x, w, ln=None, bias=None, res=None):
    x = x.cpu(); w = w.cpu()
    if ln is not None:
        x = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), ln[0].cpu(), ln[1].cpu(), 1e-5).permute(0, 3, 1, 2)
    y = F.conv2d(x, w[:, :, None, None], None if bias is None else bias.cpu())
    if res is not None: y = y + res.cpu()
    return y.to(dev)
cases = [  # K, M, H, W, ln, x2mode
    (40, 40, 64, 64, True, 0), (40, 320, 64, 64, True, 0), (40, 10, 64, 64, False, 0), (80, 80, 32, 32, True, 0), (80, 640, 32, 32, True, 0),
    (160, 160, 16, 16, True, 0), (160, 1280, 16, 16, True, 0), (160, 40, 64, 64, False, 0), (320, 80, 32, 32, False, 0), (640, 160, 16, 16, False, 0),
    (320, 40, 16, 16, True, 0), (40, 40, 64, 64, False, 1), (80, 40, 64, 64, False, 2), (80, 14, 32, 32, False, 0), (40, 40, 7, 9, True, 0),
    (40, 40, 224, 320, True, 0), (40, 320, 224, 320, True, 0), (160, 40, 224, 320, False, 0), (40, 10, 224, 320, False, 0),
    (80, 640, 112, 160, True, 0), (320, 80, 112, 160, False, 0), (160, 1280, 56, 80, True, 0), (640, 160, 56, 80, False, 0), (80, 40, 224, 320, False, 2)]
if len(sys.argv) > 1: cases = cases[15:]
for K, M, H, W, ln, mode in cases:
    B = 1 if H * W > 20000 else 2
    x = torch.randn(B, K if mode != 2 else K // 2, H, W, device=dev) * 1.7 + 0.3
    x2 = torch.randn_like(x) if mode else None
    xin = x if mode == 0 else (x + x2 if mode == 1 else torch.cat([x, x2], 1))
    w = torch.randn(M, K, device=dev) * K ** -0.5
    lnp = (torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1) if ln else None
    bias = torch.randn(M, device=dev); res = torch.randn(B, M, H, W, device=dev)
    r64 = ref64(xin, w, lnp, bias, res); r32 = ref32(xin, w, lnp, bias, res)
    e32 = (r32.double() - r64).abs().mean().item()
    out = [f"K={K:4d} M={M:5d} L={H*W:5d} ln={int(ln)} mode={mode}: torch-f32 {e32:.2e}"]
    for name, x6 in (("f32mfma", False), ("x6", True)):
        y = ops.pw_gemm(x, ops.pack_pw_weight(w, x6=x6), M, x2=x2, in_mode=mode, ln=lnp, bias=bias, res=res)
        d = y.double() - r64
        out.append(f"{name} mean {d.abs().mean().item():.2e} max {d.abs().max().item():.2e} bias {d.mean().item():+.1e}")
    print(" | ".join(out))
