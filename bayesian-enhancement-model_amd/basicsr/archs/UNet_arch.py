from bem.archs import BasicBlock, DualUpSample, Network, PatchMerging, SubNetwork  # noqa: F401
