from bem.archs import DecompSingleBranchDD  # noqa: F401
