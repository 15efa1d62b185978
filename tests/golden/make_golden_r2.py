#!/usr/bin/env python3
"""Round-2 golden vectors, generated like make_golden.py by running the REFERENCE ITSELF on CPU (build container only):

  g5_decomp23.npz  QD model2 (dilated branch convs, basicsr/QD/model2.py:154-241) and model3 (mini U-Net, eval-mode dropout =
                   identity, basicsr/QD/model3.py:166-273) with the shipped weights: wavelet-domain Q1_w / Q2_w of the
                   DDWavelet arch's MyDecomp (DecompDualBranchDDWavelet_arch.py:55-143) and the full-resolution Q1 / Q2.
  g10_train.npz    the Stage-II training step of g6_ddw's net and inputs for two steps with the reference's own modules and
                   torch.optim.AdamW / clip_grad_norm_ (image_enhancer_model.py:165-216, L1 only): per-step loss and gradient
                   norm, EVERY gradient of step 1, the parameters after step 2.
Data only (inputs, expected outputs); no reference code is copied."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from make_golden import save, synth  # noqa: E402


def main():
    ns = rh.load()
    import importlib
    torch.set_num_threads(8)
    m2, m3 = importlib.import_module("basicsr.QD.model2"), importlib.import_module("basicsr.QD.model3")
    lq, _ = synth((1, 3, 32, 40), 5)
    arrs = dict(img=lq)
    with rh.ref_ctor_env():
        for tag, mod in (("model2", m2), ("model3", m3)):
            my = ns.ddw.create_my_decomp(tag)
            d = mod.Decomp(use_wavelets=True)
            d.load_state_dict(torch.load(f"basicsr/QD/checkpoints/{tag}_999.pth")["model_state_dict"])
            my.eval(); d.eval()
            with torch.no_grad():
                a, b = my(lq)
                c, e = d(lq)
            arrs.update({f"q1w_{tag}": a, f"q2w_{tag}": b, f"q1_{tag}": c, f"q2_{tag}": e})
    save("g5_decomp23", **arrs)

    # ---- training step, two iterations, reference modules + torch optimizer ----
    g = np.load(os.path.join(HERE, "g6_ddw.npz"))
    kw = dict(in_channels=6, out_channels=3, n_feat=16, d_state=[1, 1, 1], ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp",
              use_pixelshuffle=True, drop_path=0.0, sam=False, stage=1, num_blocks=[2, 1, 1])
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        net = ns.ddw.DecompDualBranchDDWavelet(decomp_model="model4", **kw)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("decomp.") for k in missing)
    net.train()
    x6, gt = torch.from_numpy(g["x"]), torch.from_numpy(g["gt"])
    lq = x6[:, :3]
    gen = torch.Generator().manual_seed(33)
    gt_down = torch.nn.functional.interpolate(gt, scale_factor=1 / 16, mode="bilinear") + 0.1 * torch.randn(1, 3, 4, 4, generator=gen)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    losses, norms, grads = [], [], None
    for it in range(2):
        opt.zero_grad()
        up = torch.nn.functional.interpolate(gt_down, scale_factor=16, mode="bilinear", align_corners=False)
        _, preds = net(torch.cat([lq, up], 1), mask=None)
        loss = torch.nn.functional.l1_loss(preds, gt)
        loss.backward()
        if it == 0:
            grads = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.requires_grad}
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 1.0)))
        opt.step()
        losses.append(float(loss))
    save("g10_train", lq=lq, gt=gt, gt_down=gt_down, loss=np.array(losses), grad_norm=np.array(norms), grads=grads,
         params={k: p.detach() for k, p in net.named_parameters() if p.requires_grad})
    print("done")


if __name__ == "__main__":
    main()
