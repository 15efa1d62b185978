from bem.archs import CrossFusionBlock, DecompDualBranch, SEBlock, SpatialAttention  # noqa: F401
