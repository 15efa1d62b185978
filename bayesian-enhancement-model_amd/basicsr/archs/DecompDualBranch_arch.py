from bem.archs import DecompDualBranch2  # noqa: F401
