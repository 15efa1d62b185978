"""Stage-by-stage run-to-run determinism of the decomposition net at a given plane size (debugging aid)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bayesian-enhancement-model_amd"))
from bem.pipeline import build_nets
from bem import ops
net1, net2 = build_nets(device="cuda")
D = net2.decomp
g = torch.Generator().manual_seed(1)
for (H, W) in ((448, 640), (256, 256), (448, 512)):
    x = torch.rand(1, 3, H, W, generator=g).cuda()
    res = []
    for r in range(3):
        P = D._prepared()
        d = ops.quat_dwt(x, 0)
        feat = D.conv_in(d)
        dil = D.branch_q1[2].dilation[0]
        b1, b2 = D.branch_q1[2], D.branch_q2[2]
        t1 = D.branch_q1[0](feat, relu=True)
        f1 = ops.conv2d(t1, b1.weight.detach(), b1.bias.detach(), pad=dil, dilation=dil, res1=feat)
        f2 = ops.conv2d(D.branch_q2[0](feat, relu=True), b2.weight.detach(), b2.bias.detach(), pad=dil, dilation=dil, res1=feat)
        Wp, bias = ops.attn_fold(f1, f2, P["aw"], P["fw"], P["fb"])
        fused = ops.pw_gemm(f1, Wp, 32, x2=f2, in_mode=2, bias=bias)
        out = ops.conv2d(fused, P["co_w"], P["co_b"], pad=1)
        out2 = ops.conv2d(out, P["sh_w"], P["sh_b"], pad=1, res1=out)
        res.append(dict(d=d, feat=feat, t1=t1, f1=f1, f2=f2, Wp=Wp, bias=bias, fused=fused, out=out, out2=out2))
    torch.cuda.synchronize()
    print(f"{H}x{W}: " + "  ".join(f"{k} {max((res[0][k] - r[k]).abs().max().item() for r in res[1:]):.2e}" for k in res[0]))
