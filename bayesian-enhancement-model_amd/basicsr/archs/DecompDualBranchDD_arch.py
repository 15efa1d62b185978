from bem.archs import DecompDualBranch2DD  # noqa: F401
