#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE ITSELF on CPU.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference is imported through tests/golden/ref_harness.py (SURVEY.md Appendix B recipe); its
own fallbacks are what execute: selective_scan_torch (csms6s.py:29-72) and the torch
CrossScanF/CrossMergeF (csm_triton.py:22-85).  Outputs are small .npz files (inputs, weights of
reduced-width configs, expected outputs, a few autograd gradients) -- data only, no reference code.

Also converts the shipped frozen QD decomposition weights (basicsr/QD/checkpoints/model{1,4}_999.pth,
loaded with weights_only=True) to state-dict-only safetensors for the product package.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

PKG = os.path.join(ROOT, "bayesian-enhancement-model_amd")


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{k}/{kk}"] = vv.detach().cpu().numpy() if torch.is_tensor(vv) else np.asarray(vv)
        else:
            out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, **out)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def synth(shape, seed):
    """LOL-like dark image + bright target."""
    g = torch.Generator().manual_seed(seed)
    lq = 0.25 * torch.rand(shape, generator=g)
    gt = (3.5 * lq + 0.05 * torch.randn(shape, generator=g)).clamp(0, 1)
    return lq, gt


def main():
    ns = rh.load()
    torch.set_num_threads(8)
    csms6s, csm, vm = ns.csms6s, ns.csm, ns.vmamba

    # ---- QD weights (product asset) --------------------------------------------------------
    from safetensors.torch import save_file
    ck_dir = os.path.join(PKG, "basicsr", "QD", "checkpoints")
    os.makedirs(ck_dir, exist_ok=True)
    for m in ("model1", "model2", "model3", "model4"):
        ck = torch.load(os.path.join(rh.REF, "basicsr", "QD", "checkpoints", f"{m}_999.pth"),
                        map_location="cpu", weights_only=True)
        sd = {k: v.contiguous() for k, v in ck["model_state_dict"].items()}
        save_file(sd, os.path.join(ck_dir, f"{m}_999.safetensors"))
        print(f"  {m}_999.safetensors  {sum(v.numel() for v in sd.values())} params")

    # ---- G1 selective scan: reference test distributions (test_selective_scan.py:406-441) ----
    print("G1 selective scan")
    cases = dict(a=(2, 16, 64, 1, True, True), b=(2, 40, 300, 1, True, True), c=(1, 8, 4100, 1, True, True),
                 d=(2, 16, 130, 1, False, False), e=(2, 8, 96, 3, True, True))
    for tag, (Bt, KC, L, N, has_D, has_b) in cases.items():
        g = torch.Generator().manual_seed(0)
        K = 4
        u = torch.randn(Bt, KC, L, generator=g).requires_grad_()
        delta = (0.5 * torch.rand(Bt, KC, L, generator=g)).requires_grad_()
        A = (-0.5 * torch.rand(KC, N, generator=g)).requires_grad_()
        Bm = torch.randn(Bt, K, N, L, generator=g).requires_grad_()
        Cm = torch.randn(Bt, K, N, L, generator=g).requires_grad_()
        D = torch.randn(KC, generator=g).requires_grad_() if has_D else None
        bias = (0.5 * torch.rand(KC, generator=g)).requires_grad_() if has_b else None
        y = csms6s.selective_scan_fn(u, delta, A, Bm, Cm, D, bias, True, True)
        dout = torch.randn(y.shape, generator=g)
        ins = [t for t in (u, delta, A, Bm, Cm, D, bias) if t is not None]
        grads = torch.autograd.grad(y, ins, dout)
        names = [n for n, t in zip(("du", "ddelta", "dA", "dB", "dC", "dD", "dbias"), (u, delta, A, Bm, Cm, D, bias)) if t is not None]
        arrs = dict(u=u, delta=delta, A=A, B=Bm, C=Cm, y=y, dout=dout, **dict(zip(names, grads)))
        if has_D:
            arrs["D"] = D
        if has_b:
            arrs["delta_bias"] = bias
        save(f"g1_scan_{tag}", **arrs)

    # ---- G2 cross scan / merge --------------------------------------------------------------
    print("G2 cross scan/merge")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 5, 6, 7, generator=g)
    ys = torch.randn(2, 4, 5, 6, 7, generator=g)
    save("g2_cross", x=x, xs=csm.cross_scan_fn(x, True, True, False, 0, True),
         ys=ys, y=csm.cross_merge_fn(ys, True, True, False, 0, True))

    # ---- G3 Haar / quaternion ---------------------------------------------------------------
    print("G3 haar/quaternion")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 10, 12, generator=g)
    p, q = torch.randn(2, 4, 6, 5, generator=g), torch.randn(2, 4, 6, 5, generator=g)
    save("g3_haar", x=x, dwt=ns.model4.dwt_init(x), iwt=ns.model4.iwt_init(x), p=p, q=q,
         ham=ns.quaternion.hamilton_product(p, q))

    # ---- G4 VSSBlock / SS2D at C=40, 16x12 ----------------------------------------------------
    print("G4 VSSBlock")
    torch.manual_seed(100)
    blk = vm.VSSBlock(hidden_dim=40, drop_path=0, norm_layer=vm.LayerNorm2d, channel_first=True, ssm_d_state=1,
                      ssm_ratio=1, ssm_dt_rank="auto", ssm_act_layer=torch.nn.SiLU, ssm_conv=3, ssm_conv_bias=False,
                      ssm_drop_rate=0, ssm_init="v0", forward_type="v05_noz", mlp_ratio=4,
                      mlp_act_layer=torch.nn.GELU, mlp_drop_rate=0.0, mlp_type="gdmlp", use_checkpoint=False,
                      post_norm=False)
    # perturb norms / biases away from their trivial init so every term is exercised
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            if "norm" in n or n.endswith("bias") or n.endswith("Ds"):
                p_.add_(0.1 * torch.randn(p_.shape, generator=g))
    x = torch.randn(2, 40, 16, 12, generator=g).requires_grad_()
    y_ss2d = blk.op(blk.norm(x))
    y = blk(x)
    dout = torch.randn(y.shape, generator=g)
    params = dict(blk.named_parameters())
    pick = ["op.x_proj_weight", "op.A_logs", "op.dt_projs_bias", "op.in_proj.weight", "mlp.project_out.weight", "norm.weight"]
    grads = torch.autograd.grad(y, [x] + [params[k] for k in pick], dout)
    save("g4_vssblock", sd=blk.state_dict(), x=x, y=y, y_ss2d=y_ss2d, dout=dout, dx=grads[0],
         grads={k: v for k, v in zip(pick, grads[1:])})

    # ---- G5 decomposition nets with the shipped weights --------------------------------------
    print("G5 decomp")
    lq, _ = synth((1, 3, 32, 40), 5)
    with rh.ref_ctor_env():
        my4 = ns.ddw.create_my_decomp("model4")
        d1 = ns.model1.Decomp(use_wavelets=True)
        d1.load_state_dict(torch.load("basicsr/QD/checkpoints/model1_999.pth")["model_state_dict"])
        d4 = ns.model4.Decomp(use_wavelets=True)
        d4.load_state_dict(torch.load("basicsr/QD/checkpoints/model4_999.pth")["model_state_dict"])
    d1.eval(); d4.eval()
    with torch.no_grad():
        q1w, q2w = my4(lq)
        a1, a2 = d1(lq)
        b1, b2 = d4(lq)
    save("g5_decomp", img=lq, q1w_model4=q1w, q2w_model4=q2w, q1_model1=a1, q2_model1=a2, q1_model4=b1, q2_model4=b2)

    # ---- G6 Stage-II nets, reduced width (weights saved), 1x6x64x64 + one L1 training step ----
    print("G6 stage-II nets")
    kw = dict(in_channels=6, out_channels=3, n_feat=16, d_state=[1, 1, 1], ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp",
              use_pixelshuffle=True, drop_path=0.0, sam=False, stage=1, num_blocks=[2, 1, 1])
    lq, gt = synth((1, 3, 64, 64), 287128)
    g = torch.Generator().manual_seed(6)
    cond = (torch.nn.functional.interpolate(gt, scale_factor=1 / 16, mode="bilinear") + 0.1 * torch.randn(1, 3, 4, 4, generator=g))
    cond = torch.nn.functional.interpolate(cond, scale_factor=16, mode="bilinear", align_corners=False)
    x6 = torch.cat([lq, cond], 1)
    for tag, ctor, dm in (("ddw", ns.ddw.DecompDualBranchDDWavelet, "model4"), ("single", ns.single.DecompSingleBranch, "model1")):
        torch.manual_seed(100)
        with rh.ref_ctor_env():
            net = ctor(decomp_model=dm, **kw)
        gg = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for n, p_ in net.named_parameters():
                if not n.startswith("decomp.") and ("norm" in n or n.endswith("bias")):
                    p_.add_(0.05 * torch.randn(p_.shape, generator=gg))
        net.train()   # decomp stays frozen/eval-equivalent for model1/model4 (no dropout/bn)
        out = net(x6)[-1]
        loss = (out - gt).abs().mean()
        trainable = [(n, p_) for n, p_ in net.named_parameters() if p_.requires_grad]
        grads = torch.autograd.grad(loss, [p_ for _, p_ in trainable])
        gnorm = torch.sqrt(sum((g_ ** 2).sum() for g_ in grads))
        gd = {n: g_ for (n, _), g_ in zip(trainable, grads)}
        picks = [n for n in gd if n.endswith("op.x_proj_weight") or n.endswith("op.A_logs") or n.startswith("proj")][:6]
        sd = {k: v for k, v in net.state_dict().items() if not k.startswith("decomp.")}
        save(f"g6_{tag}", sd=sd, x=x6, gt=gt, out=out, loss=loss, grad_norm=gnorm, grads={k: gd[k] for k in picks},
             keys=np.array(list(net.state_dict().keys())))

    # ---- G9 sibling Stage-II archs, reduced width, 1x6x32x32 --------------------
    print("G9 sibling archs")
    kw9 = {**kw, "n_feat": 8, "num_blocks": [1, 1, 1]}
    lq9, gt9 = synth((1, 3, 32, 32), 287128)
    x9 = torch.cat([lq9, (gt9 + 0.1 * torch.randn(1, 3, 32, 32, generator=torch.Generator().manual_seed(9))).clamp(0, 1)], 1)
    contract9 = {}
    for tag, ctor, dm in (("dualdd", ns.dd.DecompDualBranch2DD, "model4"), ("dual2", ns.dual2.DecompDualBranch2, "model1"),
                          ("singledd", ns.singledd.DecompSingleBranchDD, "model1")):
        torch.manual_seed(100)
        with rh.ref_ctor_env():
            net = ctor(decomp_model=dm, **kw9)
            full = ctor(decomp_model=dm, **{**kw, "n_feat": 40, "num_blocks": [2, 2, 2]})
        gg = torch.Generator().manual_seed(7)
        with torch.no_grad():
            for n, p_ in net.named_parameters():
                if not n.startswith("decomp.") and ("norm" in n or n.endswith("bias")):
                    p_.add_(0.05 * torch.randn(p_.shape, generator=gg))
        net.eval()
        with torch.no_grad():
            res = net(x9)
        sd = {k: v for k, v in net.state_dict().items() if not k.startswith("decomp.")}
        save(f"g9_{tag}", sd=sd, x=x9, out=res[-1], first=res[0], keys=np.array(list(net.state_dict().keys())))
        contract9[type(full).__name__] = np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in full.state_dict().items()])
    save("g9_key_contract", **contract9)

    # ---- G7 Stage-I Bayesian U-Net, reduced width ----------------------------------------------
    print("G7 stage-I BNN")
    kw1 = dict(in_channels=3, out_channels=3, n_feat=16, stage=1, num_blocks=[2, 1, 1], d_state=[1, 1, 1], ssm_ratio=1,
               mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=True)
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        n1 = ns.unet.Network(**kw1)
        ns.bayesian.convert2bnn_selective(n1, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
    gg = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n, p_ in n1.named_parameters():
            if "norm" in n or n.endswith("bias") or "up_p.1" in n or "up_b.1" in n:
                p_.add_(0.05 * torch.randn(p_.shape, generator=gg))
    n1.eval()
    xin = torch.rand(2, 3, 8, 12, generator=gg) * 0.25
    with torch.no_grad():
        ns.bayesian.set_prediction_type(n1, deterministic=True)
        y_det = n1(xin)[-1]
        ns.bayesian.set_prediction_type(n1, deterministic=False)
        torch.manual_seed(287128)
        y_sto = n1(xin)[-1]
    eps = {}
    for name, mod in n1.named_modules():
        if hasattr(mod, "eps_weight"):
            eps[name + ".weight"] = mod.eps_weight.clone()
            if getattr(mod, "bias", False):
                eps[name + ".bias"] = mod.eps_bias.clone()
    bnn_layers = [f"{name}:{type(mod).__name__}" for name, mod in n1.named_modules() if hasattr(mod, "eps_weight")]
    save("g7_network", sd=n1.state_dict(), x=xin, y_det=y_det, y_sto=y_sto, eps=eps,
         keys=np.array(list(n1.state_dict().keys())), bnn_layers=np.array(bnn_layers))

    # full-width key / shape contract (names + shapes only, no weights)
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        full1 = ns.unet.Network(**{**kw1, "n_feat": 40, "num_blocks": [2, 2, 2]})
        ns.bayesian.convert2bnn_selective(full1, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
        full2 = ns.ddw.DecompDualBranchDDWavelet(decomp_model="model4", **{**kw, "n_feat": 40, "num_blocks": [2, 2, 2]})
        full3 = ns.single.DecompSingleBranch(decomp_model="model1", **{**kw, "n_feat": 40, "num_blocks": [2, 2, 2]})
    contract = {}
    for tag, net in (("Network", full1), ("DecompDualBranchDDWavelet", full2), ("DecompSingleBranch", full3)):
        contract[tag] = np.array([f"{k}|{','.join(map(str, v.shape))}" for k, v in net.state_dict().items()])
    save("g7_key_contract", **contract)

    # ---- G8 the eval.py Monte-Carlo loop (restated; nets = reference nets), 64x64, N=4 ---------
    print("G8 eval MC loop")
    torch.manual_seed(100)
    with rh.ref_ctor_env():
        s1 = ns.unet.Network(**{**kw1, "n_feat": 8, "num_blocks": [1, 1, 1]})
        ns.bayesian.convert2bnn_selective(s1, {"sigma_init": 0.05, "decay": 0.998, "pretrain": False})
        s2 = ns.ddw.DecompDualBranchDDWavelet(decomp_model="model4", **{**kw, "n_feat": 8, "num_blocks": [1, 1, 1]})
    s1.eval(); s2.eval()
    ns.bayesian.set_prediction_type(s1, deterministic=False)
    lq, gt = synth((1, 3, 60, 52), 287128)        # not a multiple of 64 -> exercises reflect pad + crop
    pad = np.pad(lq[0].permute(1, 2, 0).numpy(), ((0, 4), (0, 12), (0, 0)), "reflect")
    pad_t = torch.from_numpy(pad).permute(2, 0, 1)[None]
    # cv2 is unavailable: INTER_LINEAR x1/16 == mean of the 2x2 centre taps (documented, unpinned)
    img_down = 0.25 * (pad_t[:, :, 7::16, 7::16] + pad_t[:, :, 8::16, 7::16] + pad_t[:, :, 7::16, 8::16] + pad_t[:, :, 8::16, 8::16])
    N = 4
    torch.manual_seed(287128)
    conds, eps_all, noises, preds, finals, psnrs = [], [], [], [], [], []
    with torch.no_grad():
        for i in range(N):
            c = torch.clamp(s1(img_down)[-1], 0, 1)
            e = {}
            for name, mod in s1.named_modules():
                if hasattr(mod, "eps_weight"):
                    e[name + ".weight"] = mod.eps_weight.clone()
                    if getattr(mod, "bias", False):
                        e[name + ".bias"] = mod.eps_bias.clone()
            eps_all.append(e)
            c = torch.clamp(c * (gt.mean(dim=(2, 3), keepdims=True) / c.mean(dim=(2, 3), keepdims=True)), 0, 1)
            nz = torch.randn_like(c)
            noises.append(nz)
            conds.append(c + nz * 0.1)
        tgt = gt[0].permute(1, 2, 0).numpy()
        for c in conds:
            up = torch.nn.functional.interpolate(c, scale_factor=16, mode="bilinear", align_corners=False)
            p = torch.clamp(s2(torch.cat([pad_t, up], 1))[-1][:, :, :60, :52], 0, 1)
            preds.append(p)
            q = p[0].permute(1, 2, 0).numpy()
            q = np.clip(q * (tgt.mean(axis=(0, 1), keepdims=True) / q.mean(axis=(0, 1), keepdims=True)), 0, 1)
            finals.append(q)
            psnrs.append(10 * np.log10(1 / np.mean((tgt - q) ** 2)))
    rel = (np.array(psnrs) / max(psnrs)).tolist()
    arrs = dict(sd1=s1.state_dict(), sd2={k: v for k, v in s2.state_dict().items() if not k.startswith("decomp.")},
                lq=lq, gt=gt, img_down=img_down, conds=torch.cat(conds), noises=torch.cat(noises),
                preds=torch.cat(preds), finals=np.stack(finals), psnr=np.array(psnrs), best=np.array(rel.index(max(rel))))
    for i, e in enumerate(eps_all):
        arrs[f"eps{i}"] = e
    save("g8_eval", **arrs)
    print("done")


if __name__ == "__main__":
    main()
