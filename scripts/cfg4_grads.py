"""Loss and every parameter gradient of ONE Stage-II training forward / backward at the BASELINE config-4 shape (option file
DecompDualBranch2DDWavelet_4.yml: full width, L1 loss), through the registry seam.  Imported by tests/test_train_gpu.py; as a program it
writes the result to a file, so that the same step can be evaluated in a process started with other dispatch switches
(BEM_WGRAD_X6=0, BEM_SCAN_BWD_ROWS=0: the f32-MFMA weight gradients and the generic scan backward).
   python scripts/cfg4_grads.py out.pt [B] [S] [seed]"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
PKG = os.path.join(ROOT, "bayesian-enhancement-model_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def compute(B=16, S=256, seed=1, images=None):
    """-> (loss float, {name: gradient on the CPU}).  ``images``: optional slice of the seeded batch (batch-additivity checks)."""
    from bem import ops
    from bem.pipeline import synthetic_pair
    from basicsr.models import build_model
    from basicsr.utils.options import parse
    opt = parse(os.path.join(PKG, "Options", "DecompDualBranch2DDWavelet_4.yml"), is_train=True)
    opt["dist"] = False
    opt["condition"]["noise_level"] = 0.0            # the condition noise is a Philox draw per call: off, so that runs are comparable
    torch.manual_seed(100)
    model = build_model(opt)
    lq, gt = synthetic_pair((B, 3, S, S), seed=seed, device="cuda")
    if images is not None:
        lq, gt = lq[images].contiguous(), gt[images].contiguous()
    model.feed_train_data(dict(lq=lq, gt=gt, gt_down=ops.resize_down(gt, 16)))
    model.optimizer_g.zero_grad()
    n = lq.shape[0]
    x = torch.empty(n, 6, S, S, device="cuda")
    ops.copy_channels(model.lq.contiguous(), x, 0)
    ops.bilinear_up(model.conds, 16, dst=x, dst_c0=3)
    _, preds = model.net_g(x, mask=None)
    loss = model.cri_pix(preds, model.gt)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.net_g.named_parameters() if p.grad is not None}
    return float(loss.detach()), grads


if __name__ == "__main__":
    out = sys.argv[1]
    B, S, seed = (int(a) for a in (sys.argv[2:5] + ["16", "256", "1"][len(sys.argv) - 2:]))
    loss, grads = compute(B, S, seed)
    torch.save({"loss": loss, "grads": grads}, out)
    print(f"cfg4_grads: B={B} S={S} loss {loss:.6f}, {len(grads)} gradients -> {out}")
