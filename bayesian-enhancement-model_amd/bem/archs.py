"""Architectures of the hot path, mirroring the reference's constructors and state-dict keys:

  Network                     basicsr/archs/UNet_arch.py:364-474          (Stage-I, Bayesian after convert2bnn)
  DecompDualBranchDDWavelet   basicsr/archs/DecompDualBranchDDWavelet_arch.py:146-369
  DecompSingleBranch          basicsr/archs/DecompSingleBranch_arch.py:53-237
  Decomp (model1 / model4)    basicsr/QD/model1.py, model4.py:167-262 (+ the wavelet-domain MyDecomp :71-132)

forward(x, mask=None) -> [x, out] like the reference.  All compute goes through bem.ops (HIP).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .modules import (Conv2dK, ConvT2x2, LayerNorm2d, PwConv2d, VSSBlock, _Cache, _need_cuda, grad_mode, make_vss_level,
                      set_module_paths)
from .native import BemNativeError

_QD_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "basicsr", "QD", "checkpoints")


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def _init_weights(m):
    # UNet_arch.py:344-351 / DecompDualBranchDDWavelet_arch.py:264-271
    if isinstance(m, nn.Linear):
        _trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.weight, 1.0)
        nn.init.constant_(m.bias, 0)


def conv_down(c):
    return Conv2dK(c, c * 2, 4, 2, 1, bias=False)


# ------------------------------------------------------------------------------------------------
# Quaternion-Retinex decomposition (frozen)
# ------------------------------------------------------------------------------------------------
class _CrossAttnParams(nn.Module):
    """SymmetricCrossAttention (QD/model4.py:81-139) parameter holder; evaluated folded (bem_attn_fold)."""

    def __init__(self, dim):
        super().__init__()
        for n in ("q1_proj", "k2_proj", "v2_proj", "q2_proj", "k1_proj", "v1_proj", "out1", "out2"):
            setattr(self, n, nn.Conv2d(dim, dim, 1))


class Decomp(nn.Module):
    """wavelet=True : MyDecomp.forward -> one (B,32,h,w) tensor laid out [Q1_w (16) | Q2_w (16)].
    wavelet=False: Decomp.forward     -> one (B,8,H,W) tensor laid out [Q1 (4) | Q2 (4)] (after IWT and,
    for model4, PostSmooth).  The even/odd channel interleave of the reference is absorbed into a
    one-time permutation of the conv_out / sharpening weights."""

    def __init__(self, model="model4", wavelet_out=True, num_filters=32):
        super().__init__()
        if model not in ("model1", "model2", "model3", "model4"):
            raise ValueError(f"Unknown decomp_model: {model}")
        self.model, self.wavelet_out = model, wavelet_out
        nf = num_filters
        self.conv_in = Conv2dK(32, nf, 3, padding=1)
        if model == "model3":
            # QD/model3.py:175-178 mini U-Net.  The wavelet-domain MyDecomp of the DDWavelet arch overrides forward() with the
            # model1 / model4 op sequence, so there these three layers hold weights but are never applied (DDWavelet_arch.py:104-107).
            self.down_conv = Conv2dK(nf, nf, 3, stride=2, padding=1)
            self.mid_conv = Conv2dK(nf, nf, 3, padding=1)
            self.up_conv = ConvT2x2(nf, nf)
        d = 2 if model == "model2" else 1        # QD/model2.py:171-181: the second convolution of each branch is dilated
        self.branch_q1 = nn.Sequential(Conv2dK(nf, nf, 3, padding=1), nn.ReLU(inplace=True), Conv2dK(nf, nf, 3, padding=d, dilation=d))
        self.branch_q2 = nn.Sequential(Conv2dK(nf, nf, 3, padding=1), nn.ReLU(inplace=True), Conv2dK(nf, nf, 3, padding=d, dilation=d))
        self.cross_attn = _CrossAttnParams(nf)
        self.fuse = nn.Conv2d(nf * 2, nf, 1)
        self.conv_out = Conv2dK(nf, 32, 3, padding=1)
        self.sharpening = Conv2dK(32, 32, 3, padding=1, bias=True)
        if model == "model4" and not wavelet_out:
            self.smooth_q1 = nn.Module(); self.smooth_q1.conv = nn.Conv2d(4, 4, 3, padding=1, groups=4, bias=True)
            self.smooth_q2 = nn.Module(); self.smooth_q2.conv = nn.Conv2d(4, 4, 3, padding=1, groups=4, bias=True)
        self._cache = _Cache()

    @classmethod
    def from_shipped(cls, model, wavelet_out):
        from safetensors.torch import load_file
        m = cls(model, wavelet_out)
        sd = load_file(os.path.join(_QD_DIR, f"{model}_999.safetensors"))
        res = m.load_state_dict(sd, strict=False)      # strict=False: MyDecomp drops smooth_q* (DDWavelet_arch.py:138)
        bad = [k for k in res.unexpected_keys if not k.startswith(("smooth_q1.", "smooth_q2."))]
        if res.missing_keys or bad:
            raise RuntimeError(f"frozen decomposition weights {model}_999: missing {res.missing_keys}, unexpected {bad}")
        m.eval()
        for p in m.parameters():
            p.requires_grad = False
        return m

    def _perm(self, dev):
        if self.wavelet_out:   # [band*8 + 2j] then [band*8 + 2j + 1]
            idx = [b * 8 + 2 * j for b in range(4) for j in range(4)] + [b * 8 + 2 * j + 1 for b in range(4) for j in range(4)]
        else:                  # within each band: q1 comps then q2 comps -> IWT output is [Q1 | Q2]
            idx = [b * 8 + q for b in range(4) for q in (0, 2, 4, 6, 1, 3, 5, 7)]
        return torch.tensor(idx, device=dev, dtype=torch.long)

    def _prepared(self):
        srcs = [self.conv_out.weight, self.conv_out.bias, self.sharpening.weight, self.sharpening.bias, self.fuse.weight]

        def prep():
            dev = self.conv_out.weight.device
            p = self._perm(dev)
            ca = self.cross_attn
            aw = torch.cat([torch.cat([getattr(ca, n).weight.detach().reshape(-1), getattr(ca, n).bias.detach().reshape(-1)])
                            for n in ("q1_proj", "k2_proj", "v2_proj", "q2_proj", "k1_proj", "v1_proj", "out1", "out2")]).contiguous()
            d = dict(co_w=self.conv_out.weight.detach()[p].contiguous(), co_b=self.conv_out.bias.detach()[p].contiguous(),
                     sh_w=self.sharpening.weight.detach()[p][:, p].contiguous(), sh_b=self.sharpening.bias.detach()[p].contiguous(),
                     aw=aw, fw=self.fuse.weight.detach().reshape(32, 64).contiguous(), fb=self.fuse.bias.detach().contiguous())
            if hasattr(self, "smooth_q1"):
                d["sm_w"] = torch.cat([self.smooth_q1.conv.weight.detach(), self.smooth_q2.conv.weight.detach()], 0).contiguous()
                d["sm_b"] = torch.cat([self.smooth_q1.conv.bias.detach(), self.smooth_q2.conv.bias.detach()], 0).contiguous()
            return d
        return self._cache.get("prep", srcs, prep)

    def forward(self, x, c0=0):
        """x (B,Ct,H,W); channels [c0, c0+3) hold the RGB image to decompose."""
        _need_cuda(x)
        P = self._prepared()
        d = ops.quat_dwt(x, c0)
        feat = self.conv_in(d)
        if self.model == "model3" and not self.wavelet_out:
            if self.training and torch.is_grad_enabled():
                raise BemNativeError("Decomp model3: its attention dropout is active in train() mode (QD/model3.py:98,127); only the eval-mode "
                                     "(identity) form is on the HIP path")
            mid = self.mid_conv(self.down_conv(feat, relu=True), relu=True)
            feat = ops.add(feat, self.up_conv._forward_nograd(mid))
        dil = self.branch_q1[2].dilation[0]
        b1, b2 = self.branch_q1[2], self.branch_q2[2]
        f1 = ops.conv2d(self.branch_q1[0](feat, relu=True), b1.weight.detach(), b1.bias.detach(), pad=dil, dilation=dil, res1=feat)
        f2 = ops.conv2d(self.branch_q2[0](feat, relu=True), b2.weight.detach(), b2.bias.detach(), pad=dil, dilation=dil, res1=feat)
        Wp, bias = ops.attn_fold(f1, f2, P["aw"], P["fw"], P["fb"])
        fused = ops.pw_gemm(f1, Wp, 32, x2=f2, in_mode=2, bias=bias)
        out = ops.conv2d(fused, P["co_w"], P["co_b"], pad=1)
        out = ops.conv2d(out, P["sh_w"], P["sh_b"], pad=1, res1=out)
        if self.wavelet_out:
            return out
        q = ops.iwt(out)                               # (B,8,H,W) = [Q1 | Q2]
        if "sm_w" in P:
            q = ops.dwconv3x3(q, P["sm_w"], P["sm_b"], mode=3)
        return q


# ------------------------------------------------------------------------------------------------
# Stage-II: DecompDualBranchDDWavelet
# ------------------------------------------------------------------------------------------------
class _Dec(nn.ModuleDict):
    pass


def _decoder(dim, nb, ds, ssm_ratio, mlp_ratio, mlp_type):
    return _Dec({"up": ConvT2x2(dim, dim // 2), "fuse": PwConv2d(dim, dim // 2, bias=False),
                 "block": make_vss_level(dim // 2, nb, ds, ssm_ratio, mlp_ratio, mlp_type)})


def _check_last_act(last_act):
    if last_act is not None:
        raise NotImplementedError("last_act: only None occurs in the shipped option files")
    return nn.Identity()


class DecompDualBranchDDWavelet(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__()
        self.stage = stage
        self.num_levels = len(num_blocks)
        if isinstance(d_state, int):
            d_state = [d_state] * self.num_levels
        if decomp_model not in ("model1", "model2", "model3", "model4"):
            raise ValueError(f"Unknown decomp_model: {decomp_model}")
        self.decomp = Decomp.from_shipped(decomp_model, wavelet_out=True)
        for br in ("Q1", "Q2"):
            fc = Conv2dK(32, n_feat, 3, 1, 1, bias=True)
            nn.init.kaiming_normal_(fc.weight, mode="fan_out", nonlinearity="linear")
            nn.init.zeros_(fc.bias)
            setattr(self, f"first_conv_{br}", fc)
            enc = nn.ModuleList()
            cur = n_feat
            for i in range(self.num_levels - 1):
                enc.append(make_vss_level(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                cur *= 2
            setattr(self, f"encoders_{br}", enc)
            setattr(self, f"down_layers_{br}", nn.ModuleList([conv_down(n_feat * (2 ** i)) for i in range(self.num_levels - 1)]))
        self.bottleneck_fuse = PwConv2d(cur * 2, cur, bias=False)
        self.bottleneck_block = make_vss_level(cur, num_blocks[-1], d_state[-1], ssm_ratio, mlp_ratio, mlp_type)
        self.bottleneck_to_Q1 = PwConv2d(cur, cur, bias=False)
        self.bottleneck_to_Q2 = PwConv2d(cur, cur, bias=False)
        for br in ("Q1", "Q2"):
            d, decs = cur, nn.ModuleList()
            for i in range(self.num_levels - 2, -1, -1):
                decs.append(_decoder(d, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                d //= 2
            setattr(self, f"decoders_{br}", decs)
            pj = Conv2dK(n_feat, 16, 3, 1, 1, bias=True)
            nn.init.zeros_(pj.bias)
            setattr(self, f"proj_{br}", pj)
        self.last_act = _check_last_act(last_act)
        self.apply(_init_weights)
        # the registration order above differs from the reference only inside this constructor;
        # state-dict KEYS and shapes are identical (tests/test_host_contract.py)

    # -- pieces reused by the Monte-Carlo pipeline (decomp(img) hoisted out of the sample loop) --
    def decompose(self, x, c0):
        return self.decomp(x, c0)

    def forward_decomposed(self, d_img, d_cond, img_index=None):
        """d_img (Bi,32,h,w), d_cond (B,32,h,w); img_index: None (Bi == B) or samples-per-image count
        (image i serves batch rows [i*n, (i+1)*n))."""
        B, _, h, w = d_cond.shape
        spi = 1 if img_index is None else int(img_index)
        if d_img.shape[0] * spi != B:
            raise ValueError("forward_decomposed: image / sample batch mismatch")
        feats, skips = {}, {}
        for bi, br in enumerate(("Q1", "Q2")):
            q = torch.empty(B, 32, h, w, device=d_cond.device, dtype=d_cond.dtype)
            if spi == 1:
                ops.copy_channels(d_img, q, 0, src_c0=16 * bi, C=16)
            else:
                ops.copy_channels_rep(d_img, q, 0, spi, src_c0=16 * bi, C=16)
            ops.copy_channels(d_cond, q, 16, src_c0=16 * bi, C=16)
            f = getattr(self, f"first_conv_{br}")(q)
            sk = []
            for i in range(self.num_levels - 1):
                f = getattr(self, f"encoders_{br}")[i](f)
                f, s_ = ag.fork(f)          # two consumers (down layer, decoder skip): their gradients meet in bem_add_f32
                sk.append(s_)
                f = getattr(self, f"down_layers_{br}")[i](f)
            feats[br], skips[br] = f, sk
        fused = self.bottleneck_fuse(feats["Q1"], x2=feats["Q2"], in_mode=2)
        fused = self.bottleneck_block(fused)
        outs = []
        for br, fz in zip(("Q1", "Q2"), ag.fork(fused)):
            f = getattr(self, f"bottleneck_to_{br}")(fz)
            for j, dec in enumerate(getattr(self, f"decoders_{br}")):
                f = dec["up"](f)
                f = dec["fuse"](f, x2=skips[br][self.num_levels - 2 - j], in_mode=2)
                f = dec["block"](f)
            outs.append(getattr(self, f"proj_{br}")(f))
        if outs[0].requires_grad or outs[1].requires_grad:
            return ag.IwtHamiltonFn.apply(outs[0], outs[1])
        return ops.iwt_hamilton(outs[0], outs[1])

    def forward(self, x, mask=None):
        """Inference: kernels only, nothing recorded.  ``train()`` mode with autograd enabled (image_enhancer_model.py:165-216):
        the frozen decomposition still runs without a graph (DDWavelet_arch.py:307), the U-Nets record bem.autograd nodes."""
        _need_cuda(x)
        train = grad_mode(self)
        with torch.no_grad():
            x = x.contiguous()
            if x.shape[1] != 6:
                raise ValueError("DecompDualBranchDDWavelet expects 6 input channels (image || condition)")
            d_img, d_cond = self.decomp(x, 0), self.decomp(x, 3)
        with torch.enable_grad() if train else torch.no_grad():
            out = self.forward_decomposed(d_img, d_cond)
        return [x, out]


# ------------------------------------------------------------------------------------------------
# Sibling Stage-II archs: same blocks and kernels, different wiring of the condition
# ------------------------------------------------------------------------------------------------
class _DualBranchFullRes(nn.Module):
    """Shared body of DecompDualBranch2DD / DecompDualBranch2: two full-resolution U-Nets over quaternion maps,
    shared bottleneck, Hamilton product of the two 4-channel outputs (no wavelet stage)."""

    def _build(self, in_branch, n_feat, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type, last_act, decomp_model):
        self.num_levels = len(num_blocks)
        if isinstance(d_state, int):
            d_state = [d_state] * self.num_levels
        if decomp_model not in ("model1", "model2", "model3", "model4"):
            raise ValueError(f"Unknown decomp_model: {decomp_model}")
        self.decomp = Decomp.from_shipped(decomp_model, wavelet_out=False)
        for br in ("Q1", "Q2"):
            fc = Conv2dK(in_branch, n_feat, 3, 1, 1, bias=True)
            nn.init.kaiming_normal_(fc.weight, mode="fan_out", nonlinearity="linear")
            nn.init.zeros_(fc.bias)
            setattr(self, f"first_conv_{br}", fc)
            enc, cur = nn.ModuleList(), n_feat
            for i in range(self.num_levels - 1):
                enc.append(make_vss_level(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                cur *= 2
            setattr(self, f"encoders_{br}", enc)
            setattr(self, f"down_layers_{br}", nn.ModuleList([conv_down(n_feat * (2 ** i)) for i in range(self.num_levels - 1)]))
        self.bottleneck_fuse = PwConv2d(cur * 2, cur, bias=False)
        self.bottleneck_block = make_vss_level(cur, num_blocks[-1], d_state[-1], ssm_ratio, mlp_ratio, mlp_type)
        self.bottleneck_to_Q1 = PwConv2d(cur, cur, bias=False)
        self.bottleneck_to_Q2 = PwConv2d(cur, cur, bias=False)
        for br in ("Q1", "Q2"):
            d, decs = cur, nn.ModuleList()
            for i in range(self.num_levels - 2, -1, -1):
                decs.append(_decoder(d, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                d //= 2
            setattr(self, f"decoders_{br}", decs)
            pj = Conv2dK(n_feat, 4, 3, 1, 1, bias=True)
            nn.init.zeros_(pj.bias)
            setattr(self, f"proj_{br}", pj)
        self.last_act = _check_last_act(last_act)
        self.apply(_init_weights)

    def _dual_unet(self, q1, q2):
        """q1, q2 (B,Cin,H,W) -> Hamilton(Q1_out, Q2_out)[1:]  (B,3,H,W): kernels only in inference, bem.autograd nodes in train() mode with
        autograd on (q1, q2 come from the frozen decomposition and carry no graph)."""
        train = grad_mode(self)
        with torch.enable_grad() if train else torch.no_grad():
            feats, skips = {}, {}
            for br, q in (("Q1", q1), ("Q2", q2)):
                f = getattr(self, f"first_conv_{br}")(q)
                sk = []
                for i in range(self.num_levels - 1):
                    f = getattr(self, f"encoders_{br}")[i](f)
                    f, s1 = ag.fork(f)
                    sk.append(s1)
                    f = getattr(self, f"down_layers_{br}")[i](f)
                feats[br], skips[br] = f, sk
            fused = self.bottleneck_block(self.bottleneck_fuse(feats["Q1"], x2=feats["Q2"], in_mode=2))
            outs = []
            for br, fz in zip(("Q1", "Q2"), ag.fork(fused)):
                f = getattr(self, f"bottleneck_to_{br}")(fz)
                for j, dec in enumerate(getattr(self, f"decoders_{br}")):
                    f = dec["up"](f)
                    f = dec["fuse"](f, x2=skips[br][self.num_levels - 2 - j], in_mode=2)
                    f = dec["block"](f)
                outs.append(getattr(self, f"proj_{br}")(f))
            if train:
                return ag.HamiltonFn.apply(outs[0], outs[1])
            B, _, H, W = q1.shape
            out8 = torch.empty(B, 8, H, W, device=q1.device, dtype=q1.dtype)
            ops.copy_channels(outs[0], out8, 0)
            ops.copy_channels(outs[1], out8, 4)
            return ops.hamilton(out8)


class DecompDualBranch2DD(_DualBranchFullRes):
    """basicsr/archs/DecompDualBranchDD_arch.py:53-302: Q = cat(Q_img, Q_cond) (8 channels per branch)."""

    def __init__(self, in_channels=3, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__()
        self.stage = stage
        self._build(8, n_feat, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type, last_act, decomp_model)

    def forward(self, x, mask=None):
        _need_cuda(x)
        with torch.no_grad():
            x = x.contiguous()
            B, _, H, W = x.shape
            qi, qc = self.decomp(x, 0), self.decomp(x, 3)              # each (B,8,H,W) = [Q1 | Q2]
            qs = []
            for bi in range(2):
                q = torch.empty(B, 8, H, W, device=x.device, dtype=x.dtype)
                ops.copy_channels(qi, q, 0, src_c0=4 * bi, C=4)
                ops.copy_channels(qc, q, 4, src_c0=4 * bi, C=4)
                qs.append(q)
        return [x, self._dual_unet(qs[0], qs[1])]


class DecompDualBranch2(_DualBranchFullRes):
    """basicsr/archs/DecompDualBranch_arch.py:51-298: Q = Q_img + [cond, 0] (4 channels per branch); returns
    [x[:, 0:3], out] like the reference."""

    def __init__(self, in_channels=3, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__()
        self.stage = stage
        self._build(4, n_feat, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type, last_act, decomp_model)

    def forward(self, x, mask=None):
        _need_cuda(x)
        with torch.no_grad():
            x = x.contiguous()
            B, _, H, W = x.shape
            qi = self.decomp(x, 0)
            qs = []
            for bi in range(2):
                q = torch.empty(B, 4, H, W, device=x.device, dtype=x.dtype)
                ops.copy_channels(qi, q, 0, src_c0=4 * bi, C=4)
                ops.add_channels(x, q, 0, src_c0=3, C=3)               # + [cond, 0]
                qs.append(q)
        return [x[:, 0:3], self._dual_unet(qs[0], qs[1])]


# ------------------------------------------------------------------------------------------------
# Stage-II: DecompDualBranch (basicsr/archs/DecompModel_arch.py:101-366) -- two U-Nets, one cross-fusion, SE + spatial attention
# ------------------------------------------------------------------------------------------------
class CrossFusionBlock(nn.Module):
    """DecompModel_arch.py:57-66: x_tgt + gate * transform(x_src).  The gate is folded into the 1x1 weights and bias (row scaling, cached per
    weight version), so the block is one limb GEMM with x_tgt as its residual."""

    def __init__(self, ch):
        super().__init__()
        self.transform = PwConv2d(ch, ch, bias=True)
        self.gate = nn.Parameter(torch.ones(1, ch, 1, 1))
        self._cache = _Cache()

    def forward(self, x_src, x_tgt):
        _need_cuda(x_src)
        if grad_mode(self):                      # training: the transform as its own autograd node, then x_tgt + gate * t
            return ag.GateAddFn.apply(self.transform(x_src), self.gate, x_tgt)
        t, C = self.transform, self.transform.out_channels
        Wp, b = self._cache.get("gated", [t.weight, t.bias, self.gate], lambda: (
            ops.pack_pw_weight(ops.row_scale(t.weight.detach().reshape(C, C).contiguous(), self.gate.detach().reshape(C).contiguous())),
            ops.row_scale(t.bias.detach().contiguous(), self.gate.detach().reshape(C).contiguous())))
        return ops.pw_gemm(x_src, Wp, C, bias=b, res=x_tgt)


class SEBlock(nn.Module):
    """DecompModel_arch.py:68-83.  ``gate(x)`` returns the (B,C) channel factors; the multiplication happens inside the spatial attention
    that follows it."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())

    def gate(self, x):
        _need_cuda(x)
        return ops.se_gate(x, self.fc[0].weight.detach(), self.fc[2].weight.detach())

    def forward(self, x):
        """Training form (the scaled tensor is materialised: the backward needs it); inference goes through gate() + SpatialAttention."""
        _need_cuda(x)
        if grad_mode(self):
            return ag.SEBlockFn.apply(x, self.fc[0].weight, self.fc[2].weight)
        return ops.chan_scale(x.contiguous(), self.gate(x))


class SpatialAttention(nn.Module):
    """DecompModel_arch.py:85-99 (kernel 7, or 3)."""

    def __init__(self, kernel_size=7):
        super().__init__()
        if kernel_size not in (3, 7):
            raise ValueError("kernel size must be 3 or 7")
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)

    def forward(self, x, chan_scale=None):
        _need_cuda(x)
        if grad_mode(self):
            if chan_scale is not None:
                raise BemNativeError("SpatialAttention: the training form takes the SE-scaled tensor (SEBlock.forward), not a gate")
            return ag.SpatialAttnFn.apply(x, self.conv.weight)
        return ops.spatial_attention(x, self.conv.weight.detach(), chan_scale)


class DecompDualBranch(nn.Module):
    """DecompModel_arch.py:101-366: the image's two quaternion maps (4 channels each; the condition half of the 6-channel input is not
    read, :294) through two U-Nets with their own bottlenecks.  State-dict keys follow the reference: branch 1 without suffix, branch 2
    with the suffix ``2``."""

    def __init__(self, in_channels=3, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__()
        self.stage = stage
        self.num_levels = len(num_blocks)
        if isinstance(d_state, int):
            d_state = [d_state] * self.num_levels
        if decomp_model not in ("model1", "model2", "model3", "model4"):
            raise ValueError(f"Unknown decomp_model: {decomp_model}")
        self.decomp = Decomp.from_shipped(decomp_model, wavelet_out=False)
        for s_ in ("", "2"):
            fc = Conv2dK(4, n_feat, 3, 1, 1, bias=True)
            nn.init.kaiming_normal_(fc.weight, mode="fan_out", nonlinearity="linear")
            nn.init.zeros_(fc.bias)
            setattr(self, "first_conv" + s_, fc)
            enc, cur = nn.ModuleList(), n_feat
            for i in range(self.num_levels - 1):
                enc.append(make_vss_level(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                cur *= 2
            setattr(self, "encoders" + s_, enc)
            setattr(self, "bottleneck" + s_, make_vss_level(cur, num_blocks[-1], d_state[-1], ssm_ratio, mlp_ratio, mlp_type))
            d, decs = cur, nn.ModuleList()
            for i in range(self.num_levels - 2, -1, -1):
                decs.append(_decoder(d, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
                d //= 2
            setattr(self, "decoders" + s_, decs)
            pj = Conv2dK(n_feat, 4, 3, 1, 1, bias=True)
            nn.init.zeros_(pj.bias)
            setattr(self, "proj" + s_, pj)
            setattr(self, "down_layers" + s_, nn.ModuleList([conv_down(n_feat * (2 ** i)) for i in range(self.num_levels - 1)]))
        self.last_act = _check_last_act(last_act)
        self.cross_fusion_12, self.cross_fusion_21 = CrossFusionBlock(cur), CrossFusionBlock(cur)
        self.bottleneck_se, self.bottleneck_se2 = SEBlock(cur), SEBlock(cur)
        self.spatial_attention, self.spatial_attention2 = SpatialAttention(), SpatialAttention()
        self.apply(_init_weights)

    def forward(self, x, mask=None):
        """Inference: kernels only.  ``train()`` mode with autograd enabled (image_enhancer_model.py:165-216): the frozen decomposition runs
        without a graph, the two U-Nets, the cross-fusions, SE / attention blocks and the Hamilton product record bem.autograd nodes."""
        _need_cuda(x)
        train = grad_mode(self)
        with torch.no_grad():
            x = x.contiguous()
            B, _, H, W = x.shape
            qi = self.decomp(x, 0)                                      # (B,8,H,W) = [Q1 | Q2] of the image channels
        with torch.enable_grad() if train else torch.no_grad():
            feats, skips = [], []
            for bi, s_ in enumerate(("", "2")):
                f = getattr(self, "first_conv" + s_)(qi, cin_slice=(4 * bi, 4))
                sk = []
                for i in range(self.num_levels - 1):
                    f = getattr(self, "encoders" + s_)[i](f)
                    f, s1 = ag.fork(f)                                  # two consumers: the down layer and the decoder's skip
                    sk.append(s1)
                    f = getattr(self, "down_layers" + s_)[i](f)
                feats.append(f); skips.append(sk)
            fa, fb = ag.fork(feats[0])                                  # branch 1's deepest features feed both cross-fusions
            f2 = self.cross_fusion_12(fa, feats[1])                     # branch 2 takes from branch 1 first ...
            f2a, f2b = ag.fork(f2)
            f1 = self.cross_fusion_21(f2a, fb)                          # ... and branch 1 from the fused branch 2 (:311-312)
            outs = []
            for bi, (s_, f) in enumerate((("", f1), ("2", f2b))):
                f = getattr(self, "bottleneck" + s_)(f)
                se, sa = getattr(self, "bottleneck_se" + s_), getattr(self, "spatial_attention" + s_)
                f = sa(se(f)) if train else sa(f, chan_scale=se.gate(f))
                for j, dec in enumerate(getattr(self, "decoders" + s_)):
                    f = dec["up"](f)
                    f = dec["fuse"](f, x2=skips[bi][self.num_levels - 2 - j], in_mode=2)
                    f = dec["block"](f)
                outs.append(getattr(self, "proj" + s_)(f))
            if train:
                out = ag.HamiltonFn.apply(outs[0], outs[1])
            else:
                out8 = torch.empty(B, 8, H, W, device=x.device, dtype=x.dtype)
                ops.copy_channels(outs[0], out8, 0)
                ops.copy_channels(outs[1], out8, 4)
                out = ops.hamilton(out8)
        return [x, out]


# ------------------------------------------------------------------------------------------------
# Stage-II: DecompSingleBranch (BASELINE config 1)
# ------------------------------------------------------------------------------------------------
class DecompSingleBranch(nn.Module):
    def __init__(self, in_channels=6, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__()
        self.stage = stage
        self.num_levels = len(num_blocks)
        if isinstance(d_state, int):
            d_state = [d_state] * self.num_levels
        if decomp_model not in ("model1", "model2", "model3", "model4"):
            raise ValueError(f"Unknown decomp_model: {decomp_model}")
        self.decomp = Decomp.from_shipped(decomp_model, wavelet_out=False)
        self.conditioning_channels = 3
        self.first_conv = Conv2dK(11, n_feat, 3, 1, 1, bias=True)
        nn.init.kaiming_normal_(self.first_conv.weight, mode="fan_out", nonlinearity="linear")
        nn.init.zeros_(self.first_conv.bias)
        self.encoders = nn.ModuleList()
        cur = n_feat
        for i in range(self.num_levels - 1):
            self.encoders.append(make_vss_level(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
            cur *= 2
        self.bottleneck = make_vss_level(cur, num_blocks[-1], d_state[-1], ssm_ratio, mlp_ratio, mlp_type)
        self.decoders = nn.ModuleList()
        for i in range(self.num_levels - 2, -1, -1):
            self.decoders.append(_decoder(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type))
            cur //= 2
        self.proj = Conv2dK(n_feat, 8, 3, 1, 1, bias=True)
        nn.init.zeros_(self.proj.bias)
        self.last_act = _check_last_act(last_act)
        self.down_layers = nn.ModuleList([conv_down(n_feat * (2 ** i)) for i in range(self.num_levels - 1)])
        self.drop_path = nn.Identity()
        self.apply(_init_weights)

    def forward(self, x, mask=None):
        _need_cuda(x)
        with torch.no_grad():
            x = x.contiguous()
            B, _, H, W = x.shape
            q = self.decomp(x, 0)                                    # (B,8,H,W) = [Q1 | Q2]
            fea = torch.empty(B, 11, H, W, device=x.device, dtype=x.dtype)
            ops.copy_channels(q, fea, 0)
            ops.copy_channels(x, fea, 8, src_c0=3, C=3)
        return [x, self._run(fea)]

    def _run(self, fea):
        """The U-Net + Hamilton product on the assembled input: kernels only in inference, bem.autograd nodes in train() mode with autograd on
        (the frozen decomposition that produced ``fea`` never records)."""
        if grad_mode(self):
            with torch.enable_grad():
                return ag.Hamilton8Fn.apply(self._unet(fea))
        with torch.no_grad():
            return ops.hamilton(self._unet(fea))

    def _unet(self, fea):
        f = self.first_conv(fea)
        sk = []
        for i in range(self.num_levels - 1):
            f = self.encoders[i](f)
            f, s1 = ag.fork(f)                                       # two consumers (down layer, decoder skip); a no-op without a graph
            sk.append(s1)
            f = self.down_layers[i](f)
        f = self.bottleneck(f)
        for j, dec in enumerate(self.decoders):
            f = dec["up"](f)
            f = dec["fuse"](f, x2=sk[self.num_levels - 2 - j], in_mode=2)
            f = dec["block"](f)
        return self.proj(f)


class DecompSingleBranchDD(DecompSingleBranch):
    """basicsr/archs/DecompSingleBranchDD_arch.py:53-251: the condition is decomposed too; the single U-Net sees
    cat(Q1_img, Q2_img, Q1_cond, Q2_cond) (16 channels)."""

    def __init__(self, in_channels=6, out_channels=3, n_feat=40, stage=1, num_blocks=[2, 2, 2], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None, decomp_model="model1"):
        super().__init__(in_channels, out_channels, n_feat, stage, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type,
                         use_pixelshuffle, drop_path, use_illu, sam, last_act, decomp_model)
        del self.conditioning_channels
        self.first_conv = Conv2dK(16, n_feat, 3, 1, 1, bias=True)
        nn.init.kaiming_normal_(self.first_conv.weight, mode="fan_out", nonlinearity="linear")
        nn.init.zeros_(self.first_conv.bias)

    def forward(self, x, mask=None):
        _need_cuda(x)
        with torch.no_grad():
            x = x.contiguous()
            B, _, H, W = x.shape
            fea = torch.empty(B, 16, H, W, device=x.device, dtype=x.dtype)
            ops.copy_channels(self.decomp(x, 0), fea, 0)
            ops.copy_channels(self.decomp(x, 3), fea, 8)
        return [x, self._run(fea)]


# ------------------------------------------------------------------------------------------------
# Stage-I: Network (UNet_arch.py)
# ------------------------------------------------------------------------------------------------
class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = LayerNorm2d(4 * dim)
        self.reduction = PwConv2d(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        if grad_mode(self):
            ps = [p for p in (self.norm.weight, self.norm.bias, self.reduction.weight) if p.requires_grad]
            return ag.LnPwFn.apply(ag.SpaceToDepthFn.apply(x), self.norm, self.reduction, *ps)
        s = ops.space_to_depth(x)
        Wp, _ = self.reduction.gemm_weights(x.shape[0])
        return ops.pw_gemm(s, Wp, 2 * self.dim, ln=(self.norm.weight.detach(), self.norm.bias.detach()), ln_eps=self.norm.eps)


class _Shuffle(nn.Module):
    def forward(self, x):
        return ops.pixel_shuffle2(x)


class _Up2(nn.Module):
    def forward(self, x):
        return ops.bilinear_up(x, 2)


class DualUpSample(nn.Module):
    """scale_factor 2 only (the factor the U-Net uses, UNet_arch.py:283)."""

    def __init__(self, in_channels, scale_factor=2):
        super().__init__()
        if scale_factor != 2:
            raise NotImplementedError("DualUpSample: scale_factor 2 only")
        c = in_channels
        self.factor = scale_factor
        self.conv = PwConv2d(c, c // 2, bias=False)
        self.up_p = nn.Sequential(PwConv2d(c, 2 * c, bias=False), nn.PReLU(), _Shuffle(), PwConv2d(c // 2, c // 2, bias=False))
        self.up_b = nn.Sequential(PwConv2d(c, c, bias=True), nn.PReLU(), _Up2(), PwConv2d(c, c // 2, bias=False))

    def forward(self, x):
        if grad_mode(self):
            xa, xb = ag.fork(x)
            p = self.up_p[3](ag.PixelShuffle2Fn.apply(ag.PReLUFn.apply(self.up_p[0](xa), self.up_p[1].weight)))
            b = self.up_b[3](ag.BilinearUpFn.apply(ag.PReLUFn.apply(self.up_b[0](xb), self.up_b[1].weight), 2))
            return self.conv(p, x2=b, in_mode=2)
        p = self.up_p[3](self.up_p[2](self.up_p[0](x, prelu=self.up_p[1].weight.detach())))
        b = self.up_b[3](self.up_b[2](self.up_b[0](x, prelu=self.up_b[1].weight.detach())))
        return self.conv(p, x2=b, in_mode=2)


class BasicBlock(nn.Module):
    def __init__(self, dim, num_blocks=2, d_state=1, ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp", sam=False,
                 condition=False, bayesian=False):
        super().__init__()
        if sam or condition:
            raise NotImplementedError("BasicBlock: sam / condition are off in every shipped option file")
        self.bayesian, self.sam, self.condition = bayesian, sam, condition
        self.blocks = nn.ModuleList(list(make_vss_level(dim, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type)))

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


class SubNetwork(nn.Module):
    def __init__(self, dim=31, num_blocks=[2, 4, 4], d_state=1, ssm_ratio=1, mlp_ratio=4, mlp_type="gdmlp",
                 use_pixelshuffle=False, drop_path=0.0, sam=False):
        super().__init__()
        if not use_pixelshuffle:
            raise NotImplementedError("SubNetwork: use_pixelshuffle=True (PatchMerging/DualUpSample) is the shipped configuration")
        if drop_path:
            raise NotImplementedError("drop_path > 0")
        self.dim, self.level = dim, len(num_blocks) - 1
        if isinstance(d_state, int):
            d_state = [d_state] * len(num_blocks)
        self.encoder_layers = nn.ModuleList()
        self.drop_path = nn.Identity()
        cur = dim
        for i in range(self.level):
            self.encoder_layers.append(nn.ModuleList([
                BasicBlock(cur, num_blocks[i], d_state[i], ssm_ratio, mlp_ratio, mlp_type, sam, bayesian=True), PatchMerging(cur)]))
            cur *= 2
        self.bottleneck = BasicBlock(cur, num_blocks[-1], d_state[self.level], ssm_ratio, mlp_ratio, sam=sam, bayesian=True)
        self.decoder_layers = nn.ModuleList()
        for i in range(self.level):
            self.decoder_layers.append(nn.ModuleList([
                DualUpSample(cur, 2), PwConv2d(cur, cur // 2, bias=False),
                BasicBlock(cur // 2, num_blocks[self.level - 1 - i], d_state[self.level - 1 - i], ssm_ratio, mlp_ratio, sam=sam, bayesian=True)]))
            cur //= 2
        self.apply(_init_weights)

    def forward(self, x):
        """returns the decoder output WITHOUT the outer residual (added by Network through proj's linearity)."""
        fea, enc = x, []
        for blk, down in self.encoder_layers:
            fea = blk(fea)
            if grad_mode(self):
                fea, skip = ag.fork(fea)
                enc.append(skip)
            else:
                enc.append(fea)
            fea = down(fea)
        fea = self.bottleneck(fea)
        for i, (up, fusion, blk) in enumerate(self.decoder_layers):
            fea = up(fea)
            fea = fusion(fea, x2=enc[self.level - 1 - i], in_mode=2)
            fea = blk(fea)
        return fea


class Network(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, n_feat=40, stage=1, num_blocks=[1, 1, 1], d_state=1, ssm_ratio=1,
                 mlp_ratio=4, mlp_type="gdmlp", use_pixelshuffle=False, drop_path=0.0, use_illu=False, sam=False,
                 last_act=None):
        super().__init__()
        if stage != 1:
            raise NotImplementedError("Network: stage = 1 (the shipped configuration)")
        self.stage = stage
        self.mask_token = nn.Parameter(torch.zeros(1, n_feat, 1, 1))
        _trunc_normal_(self.mask_token, std=0.02)
        self.first_conv = Conv2dK(in_channels, n_feat, 3, 1, 1, bias=True)
        nn.init.kaiming_normal_(self.first_conv.weight, mode="fan_out", nonlinearity="linear")
        nn.init.zeros_(self.first_conv.bias)
        self.subnets = nn.ModuleList()
        self.proj = Conv2dK(n_feat, out_channels, 3, 1, 1, bias=True)
        nn.init.zeros_(self.proj.bias)
        self.last_act = _check_last_act(last_act)
        for _ in range(stage):
            self.subnets.append(SubNetwork(n_feat, num_blocks, d_state, ssm_ratio, mlp_ratio, mlp_type, use_pixelshuffle, drop_path, sam))

    def forward(self, x, mask=None):
        _need_cuda(x)
        from .modules import _SAMPLE_CTX, _TRAIN_STEP, SampleCtx, sampling
        ctx = _SAMPLE_CTX[0]
        if grad_mode(self):
            return self._forward_train(x, mask, ctx)
        if ctx is None:      # one weight sample per batch element, fresh Philox streams for this forward
            ctx = SampleCtx(x.shape[0], None, seed=torch.initial_seed() & 0xFFFFFFFF)
        with torch.no_grad(), sampling(ctx):
            set_module_paths(self)
            x = x.contiguous()
            bank = None
            if ctx.eps is None and ctx.nsets > 1 and os.environ.get("BEM_EVAL_SAMPLE_BANK", "1") != "0":
                from .modules import EvalSampleBank
                bank = self.__dict__.get("_eval_bank")
                if bank is None:
                    bank = self.__dict__["_eval_bank"] = EvalSampleBank(self)
                ctx.counter0 = ctx.counter
                if bank.usable(ctx):
                    bank.sample(ctx)                     # all leaves' weight sets in one launch; the first Philox forward goes leaf by leaf
            fea0 = self.first_conv(x)
            dec = self.subnets[0](fea0)
            # proj(fea0 + dec) = proj_nobias(fea0) + proj(dec)   (UNet_arch.py:361,470-472; conv is linear)
            base = ops.conv2d(fea0, self.proj.weight.detach(), None, pad=1)
            out = self.proj(dec, res1=base)
            ctx.bank = None
        return [x, out]

    def _forward_train(self, x, mask, ctx):
        """Training forward (UNet_arch.py:447-474 with module.training): one weight sample per Bayesian leaf for the whole batch
        (conv.py:100-104 draws eps once per forward), EMA prior update before the draw, optional MIM token mix, autograd nodes over the
        HIP kernels.  The returned prediction carries the anchor that folds the sampled-weight gradients into mu / rho after backward."""
        from .modules import _TRAIN_STEP, SampleCtx, sampling
        if ctx is None:
            ctx = SampleCtx(1, None, seed=torch.initial_seed() & 0xFFFFFFFF)
        if ctx.nsets != 1:
            raise BemNativeError("Network: a training forward shares one weight sample across the batch (SampleCtx(nsets=1))")
        step = ag.BayesStep()
        prev = _TRAIN_STEP[0]
        _TRAIN_STEP[0] = step
        ops.bump_weight_epoch()              # every training forward draws new weights: packed / transposed copies of the last draw are stale
        try:
            with sampling(ctx):
                set_module_paths(self)
                if os.environ.get("BEM_BAYES_BANK", "1") != "0":
                    from .modules import BayesBank
                    from .train import STEP_STATE
                    bank = self.__dict__.get("_bayes_bank")
                    if bank is None:
                        bank = self.__dict__["_bayes_bank"] = BayesBank(self)
                    if bank.ready() and bank.usable(ctx):
                        bank.sample(ctx, step, STEP_STATE[0])      # every leaf's prior EMA + draw + sample in one launch
                step.counter0 = ctx.counter if step.bank is None else 0
                fea = self.first_conv(x.contiguous())
                if mask is not None:
                    fea = ag.MaskTokenFn.apply(fea, mask, self.mask_token)
                fa, fb = ag.fork(fea)
                dec = self.subnets[0](fa)
                out = self.proj(ag.AddFn.apply(fb, dec))
                out = ag.BayesAnchorFn.apply(out, step)
                bank = self.__dict__.get("_bayes_bank")
                if bank is not None and not bank.ready() and ctx.eps is None:
                    bank.try_build()                               # this forward went leaf by leaf and recorded its draw order
        finally:
            _TRAIN_STEP[0] = prev
        return [x, out]
