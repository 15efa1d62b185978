#!/usr/bin/env python3
"""Round-2 golden vector for Stage-I training, generated like make_golden.py by running the REFERENCE ITSELF on CPU (build container):

  g11_stage1_train.npz   two iterations of ConditionGenerator.optimize_parameters (basicsr/models/condition_generator_model.py:176-218)
                         restated on the reference's own modules: Network (UNet_arch.py:364-474) after convert2bnn_selective, in
                         train() mode (threshold-EMA prior update + fresh eps per Bayesian leaf per forward, conv.py:84-112),
                         get_kl_loss (tools.py:76-84), l_total = 0.01 * l_kl / mini_batch + L1, clip_grad_norm_, torch.optim.AdamW.
                         Iteration 1 runs with a MIM mask (UNet_arch.py:463-466), iteration 2 without (optimize_parameters drops it
                         after the first scheduler period).  Recorded: initial parameters and prior buffers, inputs, the eps every
                         layer drew in each iteration, per-iteration l_kl / l_pix / gradient norm, every gradient of iteration 1,
                         parameters and priors after iteration 2.
Data only (inputs, expected outputs); no reference code is copied."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
from make_golden import save  # noqa: E402


def main():
    ns = rh.load()
    torch.set_num_threads(8)
    kw1 = dict(in_channels=3, out_channels=3, n_feat=16, stage=1, num_blocks=[2, 1, 1], d_state=[1, 1, 1], ssm_ratio=1,
               mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=True)
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        net = ns.unet.Network(**kw1)
        ns.bayesian.convert2bnn_selective(net, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
    gg = torch.Generator().manual_seed(11)
    with torch.no_grad():      # move the parameters off their initial values (the priors stay where init_parameters put them)
        for n, p in net.named_parameters():
            if "norm" in n or n.endswith("bias") or "up_p.1" in n or "up_b.1" in n or "mu_weight" in n:
                p.add_(0.03 * torch.randn(p.shape, generator=gg))
    net.train()
    bnn = {name: mod for name, mod in net.named_modules() if hasattr(mod, "eps_weight")}

    def priors():
        d = {}
        for name, mod in bnn.items():
            d[name + ".prior_mu_weight"], d[name + ".prior_rho_weight"] = mod.prior_mu_weight.clone(), mod.prior_rho_weight.clone()
            if mod.bias:
                d[name + ".prior_mu_bias"], d[name + ".prior_rho_bias"] = mod.prior_mu_bias.clone(), mod.prior_rho_bias.clone()
        return d

    B, H, W, mini_batch = 2, 8, 16, 8
    lq = torch.rand(B, 3, H, W, generator=gg) * 0.25
    gt = torch.rand(B, 3, H, W, generator=gg)
    mask = (torch.rand(B, H, W, generator=gg) < 0.4).float()
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    prior0 = priors()
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-4, betas=(0.9, 0.999))
    eps_steps, kls, pixs, norms, grads = [], [], [], [], None
    for it in range(2):
        opt.zero_grad()
        torch.manual_seed(4100 + it)
        _, preds = net(lq, mask=mask if it == 0 else None)
        eps = {}
        for name, mod in bnn.items():
            eps[name + ".weight"] = mod.eps_weight.clone()
            if mod.bias:
                eps[name + ".bias"] = mod.eps_bias.clone()
        eps_steps.append(eps)
        l_kl = ns.bayesian.get_kl_loss(net)
        l_pix = torch.nn.functional.l1_loss(preds, gt)
        l_total = 0.01 * l_kl / mini_batch + l_pix
        l_total.backward()
        if it == 0:
            grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()}
        norms.append(float(torch.nn.utils.clip_grad_norm_(params, 1.0)))
        opt.step()
        kls.append(float(l_kl)); pixs.append(float(l_pix))
    save("g11_stage1_train", sd=sd0, prior0=prior0, lq=lq, gt=gt, mask=mask, mini_batch=np.array(mini_batch),
         eps0=eps_steps[0], eps1=eps_steps[1], l_kl=np.array(kls), l_pix=np.array(pixs), grad_norm=np.array(norms), grads=grads,
         params={k: p.detach() for k, p in net.named_parameters()}, prior2=priors(), pred=preds.detach())
    print("done", kls, pixs, norms)


if __name__ == "__main__":
    main()
