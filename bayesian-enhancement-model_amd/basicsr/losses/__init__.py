"""build_loss (basicsr/losses/__init__.py) for the losses of the hot path's training step.

Only ``L1Loss`` is on the path (PerceptualLoss needs VGG19 weights that are not available offline);
it runs as one HIP reduction kernel forward and one elementwise kernel backward (bem.autograd.L1LossFn)."""
from copy import deepcopy

import torch.nn as nn

from basicsr.utils.registry import LOSS_REGISTRY
from bem import autograd as _ag

__all__ = ["build_loss", "L1Loss"]


@LOSS_REGISTRY.register()
class L1Loss(nn.Module):
    """basicsr/losses/losses.py:28-52: loss_weight * mean |pred - target| (reduction 'mean'; per-element weights unsupported)."""

    def __init__(self, loss_weight=1.0, reduction="mean"):
        super().__init__()
        if reduction != "mean":
            raise ValueError(f"Unsupported reduction mode: {reduction}. The HIP path implements 'mean' (the shipped option files).")
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is not None:
            raise NotImplementedError("L1Loss: element-wise weights are not used on the BEM path")
        return _ag.l1_loss(pred, target, self.loss_weight)


def build_loss(opt):
    opt = deepcopy(opt)
    loss_type = opt.pop("type")
    if loss_type == "PerceptualLoss":
        raise NotImplementedError("PerceptualLoss needs torchvision's pretrained VGG19 weights (vgg_arch.py:103-108), which cannot be "
                                  "fetched in this environment; remove `perceptual_opt` to train with the pixel loss only")
    return LOSS_REGISTRY.get(loss_type)(**opt)
