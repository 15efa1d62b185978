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
        new.prior_mu_weight.data.copy_(new.mu_weight.data)
        if m.bias is not None:
            new.mu_bias.data.copy_(m.bias.data)
            new.prior_mu_bias.data.copy_(m.bias.data)
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
    """Sum over the Bayesian leaves below ``m`` of KL(q || EMA prior) (tools.py:76-84; base_layer.py:26-40): a 0-dim device tensor
    whose backward accumulates into mu / rho .grad.  None when ``m`` holds no Bayesian leaf, as in the reference."""
    from bem import autograd as ag
    leaves = [layer for layer in m.modules() if hasattr(layer, "kl_terms")]
    if not leaves:
        return None
    params = [p for layer in leaves for t in layer.kl_terms() for p in t[:2]]
    return ag.KLFn.apply(leaves, *params)
