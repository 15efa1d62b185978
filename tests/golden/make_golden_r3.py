#!/usr/bin/env python3
"""Round-3 golden vector, generated like make_golden.py by running the REFERENCE ITSELF on CPU (build container):

  g12_dualbranch.npz    DecompDualBranch (basicsr/archs/DecompModel_arch.py:101-366): two U-Nets over the quaternion maps of the image, one
                        cross-fusion at the deepest encoder level (:57-66), SE block (:68-83) and 7x7 spatial attention (:85-99) after
                        each bottleneck, Hamilton product of the two 4-channel outputs.  Reduced width (n_feat 16, one block per level,
                        the shipped QD model4 decomposition), input 1x6x32x32; norm / bias / gate / SE / attention parameters moved
                        off their initial values.  Recorded: state dict (without the frozen decomposition), input, both outputs, the inputs and outputs of the
                        two cross-fusions, SE blocks and spatial attentions (forward hooks: the U-Net's LayerNorms hide most of a gate's
                        effect from the final output), the
                        state-dict key / shape contract of the full-width net.
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
    torch.set_num_threads(8)
    kw = dict(in_channels=6, out_channels=3, d_state=[1, 1, 1], ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=True,
              drop_path=0.0, sam=False, stage=1, decomp_model="model4")
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        net = ns.dualse.DecompDualBranch(n_feat=16, num_blocks=[1, 1, 1], **kw)
        full = ns.dualse.DecompDualBranch(n_feat=40, num_blocks=[2, 2, 2], **kw)
    gg = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.startswith("decomp."):
                continue
            if "norm" in n or n.endswith("bias") or n.endswith(".gate"):
                p.add_(0.05 * torch.randn(p.shape, generator=gg))
            elif "_se" in n or "spatial_attention" in n:
                p.add_(1.0 * torch.randn(p.shape, generator=gg))          # trunc_normal(0.02) SE weights would leave the gate at 0.5
    net.eval()
    lq, gt = synth((1, 3, 32, 32), 287128)
    x = torch.cat([lq, (gt + 0.1 * torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(9))).clamp(0, 1)], 1)
    taps = {}
    for name in ("cross_fusion_12", "cross_fusion_21", "bottleneck_se", "spatial_attention", "bottleneck_se2", "spatial_attention2"):
        def hook(mod, args, out, name=name):
            for i, a in enumerate(args):
                taps[f"{name}.in{i}"] = a.detach().clone()
            taps[f"{name}.out"] = out.detach().clone()
        getattr(net, name).register_forward_hook(hook)
    with torch.no_grad():
        res = net(x)
    sd = {k: v for k, v in net.state_dict().items() if not k.startswith("decomp.")}
    save("g12_dualbranch", sd=sd, x=x, out=res[-1], first=res[0], keys=np.array(list(net.state_dict().keys())), taps=taps,
         contract=np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in full.state_dict().items()]))


if __name__ == "__main__":
    main()
