from bem.archs import DecompSingleBranch  # noqa: F401
