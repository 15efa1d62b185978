"""Import the read-only reference (/root/reference) on CPU for fixture generation / live checks.

Test infrastructure only: used by tests/golden/make_golden.py and by the CPU-only
"live reference" tests (skipped when /root/reference is absent, e.g. on the GPU box).
Nothing here is imported by the product package, bench.py's timed path or smoke().

Recipe: namespace-package stand-ins for the heavy ``basicsr``
``__init__`` files (they pull cv2 / torchvision / lmdb), identity stubs for
``timm.models.layers`` and ``fvcore.nn`` (import-time only dependencies of
basicsr/vmamba/models/vmamba.py:13-14), cwd=/root/reference for the relative QD
checkpoint paths (DecompDualBranchDDWavelet_arch.py:58-67) and a torch.load wrapper
forcing map_location='cpu', weights_only=True (the QD checkpoints were saved from CUDA).
"""
import contextlib
import logging
import os
import sys
import types

REF = os.environ.get("BEM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "basicsr", "vmamba", "models"))


_loaded = {}


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference hot-path symbols (imported from their real files)."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF}")
    import torch
    import torch.nn as nn

    # refuse to clobber an already imported (our own) basicsr mirror
    for k in list(sys.modules):
        if k == "basicsr" or k.startswith("basicsr."):
            del sys.modules[k]
    sys.path.insert(0, REF)
    b = os.path.join(REF, "basicsr")
    # basicsr/bayesian/tools.py:1 does a bare ``import bayesian``; the reference only works because
    # UNet_arch.py:3-5 appends basicsr/ to sys.path -- reproduce that here.
    sys.path.append(b)
    _ns("basicsr", b)
    u = _ns("basicsr.utils", os.path.join(b, "utils"))
    u.get_root_logger = lambda *a, **k: logging.getLogger("basicsr")

    def scandir(dir_path, suffix=None, recursive=False, full_path=False):
        for e in sorted(os.listdir(dir_path)):
            if suffix is None or e.endswith(suffix):
                yield os.path.join(dir_path, e) if full_path else e
    u.scandir = scandir
    _ns("basicsr.archs", os.path.join(b, "archs"))
    _ns("basicsr.QD", os.path.join(b, "QD"))
    _ns("basicsr.vmamba", os.path.join(b, "vmamba"))
    _ns("basicsr.vmamba.models", os.path.join(b, "vmamba", "models"))

    # import-time-only third-party stubs
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, *a, **k):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            return x
    timm = types.ModuleType("timm"); timm.__path__ = []
    tm = types.ModuleType("timm.models"); tm.__path__ = []
    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = DropPath
    tl.trunc_normal_ = nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    fv = types.ModuleType("fvcore"); fv.__path__ = []
    fn = types.ModuleType("fvcore.nn")
    for n in ("FlopCountAnalysis", "flop_count_str", "flop_count", "parameter_count"):
        setattr(fn, n, lambda *a, **k: None)
    sys.modules.update({"fvcore": fv, "fvcore.nn": fn})

    import importlib
    with _quiet():
        ns = types.SimpleNamespace()
        ns.registry = importlib.import_module("basicsr.utils.registry")
        ns.vmamba = importlib.import_module("basicsr.vmamba.models.vmamba")
        ns.csms6s = importlib.import_module("basicsr.vmamba.models.csms6s")
        ns.csm = importlib.import_module("basicsr.vmamba.models.csm_triton")
        ns.bayesian = importlib.import_module("basicsr.bayesian")
        ns.model4 = importlib.import_module("basicsr.QD.model4")
        ns.model1 = importlib.import_module("basicsr.QD.model1")
        ns.quaternion = importlib.import_module("basicsr.QD.quaternion")
        # arch_util imports torchvision-free pieces only lazily; stub the one symbol the archs import
        au = types.ModuleType("basicsr.archs.arch_util")
        au.SAM = type("SAM", (nn.Module,), {})
        sys.modules["basicsr.archs.arch_util"] = au
        ns.unet = importlib.import_module("basicsr.archs.UNet_arch")
        ns.ddw = importlib.import_module("basicsr.archs.DecompDualBranchDDWavelet_arch")
        ns.single = importlib.import_module("basicsr.archs.DecompSingleBranch_arch")
        ns.dd = importlib.import_module("basicsr.archs.DecompDualBranchDD_arch")
        ns.dual2 = importlib.import_module("basicsr.archs.DecompDualBranch_arch")
        ns.singledd = importlib.import_module("basicsr.archs.DecompSingleBranchDD_arch")
        ns.dualse = importlib.import_module("basicsr.archs.DecompModel_arch")
    ns.torch = torch
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def _quiet():
    import io
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(io.StringIO()):
            yield


@contextlib.contextmanager
def ref_ctor_env():
    """cwd=/root/reference and CPU-safe torch.load while a reference arch constructor runs."""
    import torch
    old_cwd = os.getcwd()
    old_load = torch.load

    def cpu_load(f, *a, **k):
        k["map_location"] = "cpu"
        k["weights_only"] = True
        return old_load(f, *a, **k)
    os.chdir(REF)
    torch.load = cpu_load
    try:
        with _quiet():
            yield
    finally:
        torch.load = old_load
        os.chdir(old_cwd)
