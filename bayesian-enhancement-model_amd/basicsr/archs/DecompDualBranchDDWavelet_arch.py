from bem.archs import DecompDualBranchDDWavelet  # noqa: F401
