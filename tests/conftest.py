import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bayesian-enhancement-model_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Returns {key: torch tensor or numpy array}; nested 'a/b' keys become dict a -> {b: ...}."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        v = torch.from_numpy(v) if v.dtype.kind == "f" else v
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def qd_state_dict(model, prefix="decomp."):
    from safetensors.torch import load_file
    sd = load_file(os.path.join(PKG, "basicsr", "QD", "checkpoints", f"{model}_999.safetensors"))
    return {prefix + k: v for k, v in sd.items()}
