"""BNN conversion / prediction-mode helpers (basicsr/bayesian/tools.py:48-84)."""
from bem.modules import (Conv2dReparameterization, DwConv2d, Linear2d, Linear2dReparameterization,  # noqa: F401
                         PwConv2d)


def _convert_leaf(m, cfg):
    if isinstance(m, Linear2d):
        new = Linear2dReparameterization(m.in_features, m.out_features, bias=m.bias is not None,
                                         decay=cfg["decay"], sigma_init=cfg["sigma_init"])
    elif isinstance(m, (PwConv2d, DwConv2d)):
        new = Conv2dReparameterization(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation,
                                       m.groups, bias=m.bias is not None, decay=cfg["decay"], sigma_init=cfg["sigma_init"])
    else:
        return None
    if cfg.get("pretrain"):
        new.mu_weight.data.copy_(m.weight.data.view_as(new.mu_weight))
        if m.bias is not None:
            new.mu_bias.data.copy_(m.bias.data)
    return new.to(m.weight.device)


def convert2bnn(m, config):
    """Replace every Linear*/Conv* leaf below ``m`` by its <Class>Reparameterization twin (tools.py:53-63)."""
    for name, child in list(m._modules.items()):
        if child is None:
            continue
        if child._modules:
            convert2bnn(child, config)
        else:
            new = _convert_leaf(child, config)
            if new is not None:
                setattr(m, name, new)


def convert2bnn_selective(model, config):
    """Only under modules tagged ``.bayesian = True`` (tools.py:48-51)."""
    for _, module in list(model.named_modules()):
        if getattr(module, "bayesian", False):
            convert2bnn(module, config)


def set_prediction_type(model, deterministic=True):
    for _, module in model.named_modules():
        if hasattr(module, "deterministic"):
            module.deterministic = bool(deterministic)


def get_kl_loss(m):
    raise NotImplementedError("KL / EMA-prior training of the Bayesian layers is a later row of SURVEY.md section 8f")
