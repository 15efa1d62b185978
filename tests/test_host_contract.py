"""CPU: the C-ABI library loads and exports every symbol include/bem_hip.h declares; the basicsr-compatible
host layer (registries, option parsing, state-dict key contract, BNN conversion, error behaviour)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG, ROOT


@pytest.fixture(scope="module")
def native():
    from bem import native as n
    if not os.path.exists(n.LIB_PATH):
        n.build()
    return n


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "bem_hip.h")).read()
    declared = set(re.findall(r"\b(bem_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("bem_pw_args")
    assert len(declared) >= 25
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in bem_hip.h but not exported by libbem_hip.so"
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    assert lib.bem_abi_version() == 1
    assert lib.bem_pw_packed_elems(40, 40) == 2 * 20 * 64


def test_rejects_null_without_gpu(native):
    """Argument validation happens before any HIP call, so it is checkable on a GPU-less host."""
    lib = native.lib()
    rc = lib.bem_dwt_f32(None, None, 1, 1, 2, 2, None)
    assert rc == 1 and b"null" in lib.bem_last_error()
    rc = lib.bem_selective_scan_fwd_f32(*([None] * 8), 1, 4, 8, 1, 1, 1, None)
    assert rc == 1


def test_no_cpu_fallback():
    from bem import ops
    from bem.native import BemNativeError
    import bem.archs as A
    with pytest.raises(BemNativeError):
        ops.dwt(torch.zeros(1, 1, 2, 2))
    net = A.DecompSingleBranch(n_feat=8, num_blocks=[1, 1, 1], decomp_model="model1")
    with pytest.raises(BemNativeError):
        net(torch.zeros(1, 6, 16, 16))


def test_registry_semantics():
    from basicsr.utils.registry import ARCH_REGISTRY, MODEL_REGISTRY, Registry
    import basicsr.archs, basicsr.models  # noqa: F401
    for n in ("Network", "DecompDualBranchDDWavelet", "DecompSingleBranch"):
        assert n in ARCH_REGISTRY
    for n in ("ConditionGenerator", "ImageEnhancer"):
        assert n in MODEL_REGISTRY
    with pytest.raises(KeyError):
        ARCH_REGISTRY.get("NoSuchArch")
    r = Registry("t")

    @r.register(suffix="basicsr")
    class Foo:
        pass
    assert r.get("Foo") is Foo                     # falls back to Foo_basicsr like the reference
    with pytest.raises(AssertionError):
        r.register(Foo, suffix="basicsr")


@pytest.mark.parametrize("yml,arch,mtype", [("CG_UNet_LOLv1.yml", "Network", "ConditionGenerator"),
                                            ("DecompDualBranch2DDWavelet_4.yml", "DecompDualBranchDDWavelet", "ImageEnhancer"),
                                            ("DecompSingleBranch_1.yml", "DecompSingleBranch", "ImageEnhancer")])
def test_parse_and_build_model_key_contract(yml, arch, mtype):
    """parse(yml) -> build_model(opt).net_g has exactly the reference's state-dict keys and shapes
    (g7_key_contract.npz was written from the reference's own constructors)."""
    from basicsr.models import build_model
    from basicsr.utils.options import parse
    opt = parse(os.path.join(PKG, "Options", yml), is_train=False)
    assert opt["model_type"] == mtype and opt["is_train"] is False and opt["name"] == yml[:-4]
    assert opt["condition"]["scale_down"] == 16 and opt["condition"]["noise_level"] == 0.1
    assert opt["datasets"]["val"]["phase"] == "val" and "results_root" in opt["path"]
    opt["num_gpu"] = 0
    net = build_model(opt).net_g
    ref = np.load(os.path.join(GOLDEN, "g7_key_contract.npz"))[arch].tolist()
    mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()]
    assert sorted(mine) == sorted(ref)


def test_dualbranch_se_attention_key_contract():
    """Options/DecompDualBranch_4.yml -> ARCH_REGISTRY 'DecompDualBranch' (basicsr/archs/DecompModel_arch.py:101): state-dict keys and
    shapes of the full-width net equal the reference's (recorded in g12_dualbranch.npz), so its checkpoints load strictly."""
    from basicsr.models import build_model
    from basicsr.utils.options import parse
    opt = parse(os.path.join(PKG, "Options", "DecompDualBranch_4.yml"), is_train=False)
    opt["num_gpu"] = 0
    net = build_model(opt).net_g
    assert type(net).__name__ == "DecompDualBranch"
    ref = np.load(os.path.join(GOLDEN, "g12_dualbranch.npz"))["contract"].tolist()
    mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()]
    assert sorted(mine) == sorted(ref)
    import basicsr.archs.DecompModel_arch as shim
    assert shim.DecompDualBranch is type(net) and shim.SEBlock and shim.SpatialAttention and shim.CrossFusionBlock


@pytest.mark.skipif(not os.path.isdir("/root/reference/Options"), reason="reference tree not present")
def test_reference_option_files_parse():
    """All 14 Decomp* option files the reference ships parse and build through the mirror, unmodified."""
    from basicsr.archs import build_network
    from basicsr.utils.options import parse
    files = sorted(f for f in os.listdir("/root/reference/Options") if f.startswith("Decomp") and f.endswith(".yml"))
    assert len(files) == 14, files
    for f in files:
        opt = parse(os.path.join("/root/reference/Options", f), is_train=False)
        net = build_network(opt["network_g"])
        assert type(net).__name__ == opt["network_g"]["type"]


def test_bnn_conversion_and_prediction_type():
    from basicsr.archs import build_network
    from basicsr.bayesian import convert2bnn_selective, set_prediction_type
    net = build_network(dict(type="Network", n_feat=8, num_blocks=[1, 1, 1], d_state=[1, 1, 1], use_pixelshuffle=True))
    n_det = sum(p.numel() for p in net.parameters())
    convert2bnn_selective(net, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
    bayes = [(n, m) for n, m in net.named_modules() if hasattr(m, "deterministic")]
    assert len(bayes) == 6 * 5                                  # 6 leaves per VSSBlock, 5 blocks
    assert {type(m).__name__ for _, m in bayes} == {"Conv2dReparameterization", "Linear2dReparameterization"}
    assert all(not m.deterministic for _, m in bayes)
    set_prediction_type(net, True)
    assert all(m.deterministic for _, m in bayes)
    assert sum(p.numel() for p in net.parameters()) > n_det
    rho = bayes[0][1].rho_weight
    assert torch.allclose(torch.log1p(torch.exp(rho)), torch.full_like(rho, 0.05), atol=1e-6)   # sigma_init
    # non-Bayesian parts stay plain (first_conv, proj, PatchMerging, DualUpSample, x_proj/dt_proj/A/D)
    keys = net.state_dict().keys()
    assert "first_conv.weight" in keys and "subnets.0.encoder_layers.0.1.reduction.weight" in keys
    assert "subnets.0.bottleneck.blocks.0.op.x_proj_weight" in keys


@pytest.mark.parametrize("yml,arch", [("DecompDualBranch2DD_4.yml", "DecompDualBranch2DD"), ("DecompDualBranch2_1.yml", "DecompDualBranch2"),
                                      ("DecompSingleBranchDD_1.yml", "DecompSingleBranchDD")])
def test_sibling_arch_key_contract(yml, arch):
    from basicsr.models import build_model
    from basicsr.utils.options import parse
    opt = parse(os.path.join(PKG, "Options", yml), is_train=False)
    opt["num_gpu"] = 0
    net = build_model(opt).net_g
    assert type(net).__name__ == arch
    ref = np.load(os.path.join(GOLDEN, "g9_key_contract.npz"))[arch].tolist()
    mine = [f"{k}|{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()]
    assert sorted(mine) == sorted(ref)


def test_product_package_never_imports_the_oracle():
    """The oracle is the checker (tests/, smoke(), bench.py's cpu_baseline): nothing under the product package may import it."""
    import re
    pkg = os.path.join(ROOT, "bayesian-enhancement-model_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, f)).read(), re.M):
                bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_native_seam_module_surface():
    """selective_scan_cuda_oflex exposes fwd / bwd with the reference's arity (selective_scan_oflex.cpp:157-165,245-254) and
    rejects CPU tensors with RuntimeError (TORCH_CHECK parity) before touching the library."""
    import inspect
    import selective_scan_cuda_oflex as ext
    assert list(inspect.signature(ext.fwd).parameters) == ["u", "delta", "A", "B", "C", "D", "delta_bias", "delta_softplus", "nrows", "out_float"]
    assert list(inspect.signature(ext.bwd).parameters) == ["u", "delta", "A", "B", "C", "D", "delta_bias", "dout", "x", "delta_softplus", "nrows"]
    u = torch.zeros(1, 4, 8)
    with pytest.raises(RuntimeError):
        ext.fwd(u, u, torch.zeros(4, 1), torch.zeros(1, 1, 1, 8), torch.zeros(1, 1, 1, 8), None, None, True, 1, True)
    from basicsr.QD.quaternion import hamilton_product  # noqa: F401  (reference name)
    from basicsr.vmamba.models.csms6s import SelectiveScanCuda, selective_scan_fn  # noqa: F401


def test_training_option_file_and_scheduler():
    """Options/DecompDualBranch2DDWavelet_4.yml parses with the reference's train schema; the cyclic cosine schedule follows
    basicsr/models/lr_scheduler.py:186-230 (position = first cycle whose cumulative end is >= the iteration)."""
    import math
    from basicsr.models.lr_scheduler import CosineAnnealingRestartCyclicLR
    from basicsr.utils.options import parse
    opt = parse(os.path.join(ROOT, "bayesian-enhancement-model_amd", "Options", "DecompDualBranch2DDWavelet_4.yml"), is_train=True)
    tr = opt["train"]
    assert tr["optim_g"] == {"type": "AdamW", "lr": 2e-4, "weight_decay": 1e-4, "betas": [0.9, 0.999]} and tr["max_grad_norm"] == 1
    assert tr["pixel_opt"]["type"] == "L1Loss" and "perceptual_opt" not in tr and opt["path"]["experiments_root"].endswith("DecompDualBranch2DDWavelet_4")
    # Stage-I: the keys ConditionGenerator.optimize_parameters reads (condition_generator_model.py:176-218)
    cg = parse(os.path.join(ROOT, "bayesian-enhancement-model_amd", "Options", "CG_UNet_LOLv1.yml"), is_train=True)
    assert cg["model_type"] == "ConditionGenerator" and cg["datasets"]["train"]["mini_batch_sizes"][0] == 8
    assert cg["train"]["scheduler"]["periods"][0] == 150000 and cg["train"]["mixing_augs"]["mixup"] is False and cg["train"]["pixel_opt"]["type"] == "L1Loss"
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.SGD([p], lr=1.0)
    s = CosineAnnealingRestartCyclicLR(o, periods=[4, 6], restart_weights=[1, 0.5], eta_mins=[0.1, 0.01])
    lrs = []
    for _ in range(10):
        lrs.append(o.param_groups[0]["lr"])
        o.step(); s.step()

    def ref(t):
        i = 0 if t <= 4 else 1
        start, per, w, lo = (0, 4, 1, 0.1) if i == 0 else (4, 6, 0.5, 0.01)
        return lo + w * 0.5 * (1.0 - lo) * (1 + math.cos(math.pi * (t - start) / per))
    assert all(abs(a - ref(t)) < 1e-12 for t, a in enumerate(lrs)), (lrs, [ref(t) for t in range(10)])


def test_checkpoint_files_and_resume_paths(tmp_path):
    """save_network / save_training_state write the reference's file formats (base_model.py:236-280,345-376), check_resume points the
    pretrain paths at the resumed iteration (misc.py:94-124) and --auto_resume picks the newest state (train.py:74-94)."""
    import torch
    from basicsr.models.base_model import BaseModel
    from basicsr.utils import check_resume, load_resume_state
    opt = {"name": "run", "num_gpu": 0, "is_train": True, "network_g": {"type": "x"},
           "path": {"models": str(tmp_path / "experiments/run/models"), "training_states": str(tmp_path / "experiments/run/training_states"),
                    "pretrain_network_g": "old.pth", "param_key_g": "params_ema", "resume_state": None}}
    m = BaseModel(opt)
    net = torch.nn.DataParallel(torch.nn.Linear(3, 2)) if False else torch.nn.Linear(3, 2)
    path = m.save_network(net, "net_g", 5)
    ck = torch.load(path, weights_only=True)
    assert path.endswith("net_g_5.pth") and set(ck) == {"params"} and set(ck["params"]) == {"weight", "bias"}
    assert m.save_network([net, net], "net_g", -1, param_key=["params", "params_ema"]).endswith("net_g_latest.pth")
    opt2 = torch.optim.AdamW(net.parameters(), lr=1e-3)
    net(torch.ones(1, 3)).sum().backward(); opt2.step()
    m.optimizers, m.schedulers = [opt2], [torch.optim.lr_scheduler.StepLR(opt2, 10)]
    for it in (5, 20, 100):
        sp = m.save_training_state(3, it, best_metric={"psnr": 21.5, "iter": it})
    st = torch.load(sp, weights_only=True)
    assert set(st) == {"epoch", "iter", "optimizers", "schedulers", "best_metric"} and st["iter"] == 100 and st["epoch"] == 3
    assert m.save_training_state(3, -1, best_metric={}) is None
    with pytest.raises(KeyError):
        m.save_training_state(3, 7)
    # auto resume: newest state, pretrain path and param key rewritten
    opt["auto_resume"] = True
    state = load_resume_state(opt, experiments_root=str(tmp_path / "experiments"))
    assert state["iter"] == 100 and opt["path"]["resume_state"].endswith("100.state")
    assert opt["path"]["pretrain_network_g"].endswith("models/net_g_100.pth") and opt["path"]["param_key_g"] == "params"
    m.resume_training(state)
    assert m.optimizers[0].state_dict()["state"][0]["step"] == 1
    # explicit resume_state path, and nothing to resume
    opt["auto_resume"] = False
    opt["path"]["resume_state"] = str(tmp_path / "experiments/run/training_states/20.state")
    assert load_resume_state(opt)["iter"] == 20
    opt["path"]["resume_state"] = None
    assert load_resume_state(opt) is None
