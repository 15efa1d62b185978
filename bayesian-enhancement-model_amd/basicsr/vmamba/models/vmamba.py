from bem.modules import LayerNorm2d, Linear2d, SS2D, VSSBlock, gdMlp  # noqa: F401
