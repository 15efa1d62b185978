"""One Stage-II training step at the bench size (B = 16, 256x256) with the x6 and with the f32-MFMA weight-gradient kernels: same loss, gradient
norm and gradients (to summation-order noise)."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "bayesian-enhancement-model_amd")
sys.path.insert(0, PKG)
from bem import ops
from bem.pipeline import synthetic_pair
from basicsr.models import build_model
from basicsr.utils.options import parse
res = {}
for form in (True, False):
    ops.WGRAD_X6 = form
    opt = parse(os.path.join(PKG, "Options", "DecompDualBranch2DDWavelet_4.yml"), is_train=True)
    opt["dist"] = False
    torch.manual_seed(100)
    model = build_model(opt)
    lq, gt = synthetic_pair((16, 3, 256, 256), seed=1, device="cuda")
    model.feed_train_data(dict(lq=lq, gt=gt, gt_down=ops.resize_down(gt, 16)))
    model.optimizer_g.zero_grad()
    # forward + backward only (no step): read raw gradients
    import torch.nn.functional as F
    x = torch.empty(16, 6, 256, 256, device="cuda"); ops.copy_channels(model.lq.contiguous(), x, 0); ops.bilinear_up(model.conds, 16, dst=x, dst_c0=3)
    _, preds = model.net_g(x, mask=None)
    loss = model.cri_pix(preds, model.gt); loss.backward()
    torch.cuda.synchronize()
    res[form] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.net_g.named_parameters() if p.grad is not None})
la, ga = res[True]; lb, gb = res[False]
print("loss x6 / f32:", la, lb)
na = sum(float((g.double() ** 2).sum()) for g in ga.values()) ** 0.5; nb = sum(float((g.double() ** 2).sum()) for g in gb.values()) ** 0.5
print("grad norm x6 / f32:", na, nb, "rel diff", abs(na - nb) / nb)
worst = max((float((ga[k] - gb[k]).abs().max()) / (float(gb[k].abs().max()) + 1e-12), k) for k in ga)
print("largest relative gradient difference:", worst)
sys.exit(0 if worst[0] < 2e-4 and abs(na - nb) / nb < 1e-5 else 1)
