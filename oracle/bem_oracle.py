"""CPU oracle: a functional PyTorch-CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg as the *checker*; never by the product package
(bayesian-enhancement-model_amd/), which has no CPU path and fails loudly without its HIP library.

Every function restates one piece of vfrantc/Bayesian-Enhancement-Model (paths relative to
/root/reference) and is pinned against golden vectors produced by importing the reference
itself on CPU (tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).  Parity status: PINNED for every function below except
``cv2_resize_down`` (cv2 is not installed anywhere in this environment; restated from the
INTER_LINEAR definition, "parity unpinned").

All tensors are float32 NCHW, weights come from a flat ``state_dict``-style mapping
``sd[prefix + name]`` using the reference's own key names.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------------------------
# A9  selective scan            basicsr/vmamba/models/csms6s.py:29-72 (selective_scan_torch)
# ----------------------------------------------------------------------------------------------

def selective_scan_ref(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """u, delta: (Bt, K*Cd, L); A: (K*Cd, N); B, C: (Bt, K, N, L); D, delta_bias: (K*Cd).

    h_t = exp(dt_t * A) * h_{t-1} + dt_t * B_t * u_t ;  y_t = <C_t, h_t> + D * u_t,
    dt = softplus(delta + delta_bias).  Time loop kept in the per-step form of csms6s.py:61-66.
    """
    Bt, K, N, L = B.shape
    KC = u.shape[1]
    Cd = KC // K
    dt = delta.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[None, :, None]
    if delta_softplus:
        dt = F.softplus(dt)
    u = u.float()
    Bx = B.float().repeat_interleave(Cd, dim=1)  # (Bt, KC, N, L): group k serves channels k*Cd..
    Cx = C.float().repeat_interleave(Cd, dim=1)
    dA = torch.exp(dt.unsqueeze(2) * A.float()[None, :, :, None])          # (Bt, KC, N, L)
    dBu = (dt * u).unsqueeze(2) * Bx                                       # (Bt, KC, N, L)
    h = torch.zeros(Bt, KC, N)
    ys = []
    for t in range(L):
        h = dA[..., t] * h + dBu[..., t]
        ys.append((h * Cx[..., t]).sum(-1))
    y = torch.stack(ys, dim=2)
    if D is not None:
        y = y + u * D.float()[None, :, None]
    return y


_C_LIB = None


def _c_lib():
    """oracle/_build/liboracle_scan.so (compiled by oracle/Makefile from selective_scan_oracle.c)."""
    global _C_LIB
    if _C_LIB is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "_build", "liboracle_scan.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", here])
        _C_LIB = ctypes.CDLL(path)
        _C_LIB.oracle_selective_scan_f32.restype = ctypes.c_int
    return _C_LIB


def selective_scan_c(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """Same contract as selective_scan_ref, evaluated by the plain-C restatement (fast enough for
    L = 16384).  The C code accumulates the recurrence in float like the reference."""
    lib = _c_lib()
    Bt, K, N, L = B.shape
    KC = u.shape[1]
    f = lambda t: None if t is None else t.detach().contiguous().float()
    u_, d_, A_, B_, C_, D_, b_ = map(f, (u, delta, A, B, C, D, delta_bias))
    out = torch.empty(Bt, KC, L, dtype=torch.float32)
    p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
    rc = lib.oracle_selective_scan_f32(p(u_), p(d_), p(A_), p(B_), p(C_), p(D_), p(b_), p(out),
                                       ctypes.c_int(Bt), ctypes.c_int(KC), ctypes.c_int(L),
                                       ctypes.c_int(N), ctypes.c_int(K), ctypes.c_int(int(delta_softplus)))
    if rc != 0:
        raise RuntimeError(f"oracle_selective_scan_f32 rc={rc}")
    return out


# ----------------------------------------------------------------------------------------------
# A8  cross scan / merge        basicsr/vmamba/models/csm_triton.py:22-85
# ----------------------------------------------------------------------------------------------

def cross_scan_ref(x):
    """(B,C,H,W) -> (B,4,C,H*W): row-major, column-major, and both reversed."""
    B, C, H, W = x.shape
    row = x.reshape(B, C, H * W)
    col = x.transpose(2, 3).reshape(B, C, H * W)
    return torch.stack([row, col, row.flip(-1), col.flip(-1)], dim=1)


def cross_merge_ref(ys):
    """(B,4,C,H,W) (each plane holding a length-H*W sequence) -> (B,C,H*W)."""
    B, K, C, H, W = ys.shape
    s = ys.reshape(B, 4, C, H * W)
    row = s[:, 0] + s[:, 2].flip(-1)
    col = s[:, 1] + s[:, 3].flip(-1)
    return row + col.reshape(B, C, W, H).transpose(2, 3).reshape(B, C, H * W)


# ----------------------------------------------------------------------------------------------
# A11 Haar / quaternion         basicsr/QD/model4.py:7-37, basicsr/QD/quaternion.py:3-17
# ----------------------------------------------------------------------------------------------

def dwt_ref(x):
    a = x[:, :, 0::2, 0::2] / 2   # even row, even col
    b = x[:, :, 1::2, 0::2] / 2   # odd row, even col
    c = x[:, :, 0::2, 1::2] / 2   # even row, odd col
    d = x[:, :, 1::2, 1::2] / 2
    return torch.cat([a + b + c + d, -a - b + c + d, -a + b - c + d, a - b - c + d], dim=1)


def iwt_ref(x):
    B, C4, H, W = x.shape
    C = C4 // 4
    ll, hl, lh, hh = (x[:, i * C:(i + 1) * C].float() / 2 for i in range(4))
    out = torch.zeros(B, C, 2 * H, 2 * W, dtype=torch.float32)   # always f32 (model4.py:31)
    out[:, :, 0::2, 0::2] = ll - hl - lh + hh
    out[:, :, 1::2, 0::2] = ll - hl + lh - hh
    out[:, :, 0::2, 1::2] = ll + hl - lh - hh
    out[:, :, 1::2, 1::2] = ll + hl + lh + hh
    return out


def hamilton_ref(p, q):
    r1, i1, j1, k1 = p.unbind(1)
    r2, i2, j2, k2 = q.unbind(1)
    return torch.stack([
        r1 * r2 - i1 * i2 - j1 * j2 - k1 * k2,
        r1 * i2 + i1 * r2 + j1 * k2 - k1 * j2,
        r1 * j2 - i1 * k2 + j1 * r2 + k1 * i2,
        r1 * k2 + i1 * j2 - j1 * i2 + k1 * r2], dim=1)


# ----------------------------------------------------------------------------------------------
# A7  VSSBlock pieces           basicsr/vmamba/models/vmamba.py:42-63,116-133,547-716,1319-1334
# ----------------------------------------------------------------------------------------------

def layernorm2d_ref(x, w, b, eps=1e-5):
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def _w4(w):
    return w if w.dim() == 4 else w[:, :, None, None]


class EpsSource:
    """Bayesian weight sampling (basicsr/bayesian/conv.py:106-114, linear.py:82-90).

    ``eps`` maps '<layer prefix>weight' / '<layer prefix>bias' to the N(0,1) draw used for that
    layer (recorded from the reference, or freshly drawn); ``None`` means deterministic (mu)."""

    def __init__(self, eps: Optional[Dict[str, torch.Tensor]] = None, generator=None, record=False):
        self.eps = eps
        self.gen = generator
        self.record = {} if record else None

    def draw(self, key, like):
        if self.eps is not None:
            return self.eps[key].reshape(like.shape)
        e = torch.randn(like.shape, generator=self.gen)
        if self.record is not None:
            self.record[key] = e
        return e


def _wb(sd: SD, pre: str, src: Optional[EpsSource]):
    """(weight, bias) of a conv/linear leaf, plain or Bayesian ('mu_weight'/'rho_weight')."""
    if pre + "weight" in sd:
        return sd[pre + "weight"], sd.get(pre + "bias")
    mu_w = sd[pre + "mu_weight"]
    mu_b = sd.get(pre + "mu_bias")
    if src is None:
        return mu_w, mu_b
    w = mu_w + torch.log1p(torch.exp(sd[pre + "rho_weight"])) * src.draw(pre + "weight", mu_w)
    b = None
    if mu_b is not None:
        b = mu_b + torch.log1p(torch.exp(sd[pre + "rho_bias"])) * src.draw(pre + "bias", mu_b)
    return w, b


def ss2d_core_ref(sd: SD, pre: str, x, scan=selective_scan_ref):
    """forward_corev2 (vmamba.py:547-698) for forward_type v05_noz: cross-scan, x_proj, dt_proj,
    selective scan in 4 directions, cross-merge, out_norm."""
    B, Cd, H, W = x.shape
    L = H * W
    xw = sd[pre + "x_proj_weight"]            # (4, R+2N, Cd)
    dtw = sd[pre + "dt_projs_weight"]         # (4, Cd, R)
    K, _, R = dtw.shape
    N = sd[pre + "A_logs"].shape[1]
    xs = cross_scan_ref(x)                                                    # (B,4,Cd,L)
    x_dbl = torch.einsum("bkcl,kjc->bkjl", xs, xw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kcr->bkcl", dts, dtw)
    A = -torch.exp(sd[pre + "A_logs"].float())
    ys = scan(xs.reshape(B, K * Cd, L), dts.reshape(B, K * Cd, L).contiguous(), A,
              Bs.contiguous(), Cs.contiguous(), sd[pre + "Ds"].float(),
              sd[pre + "dt_projs_bias"].reshape(-1).float(), True)
    y = cross_merge_ref(ys.reshape(B, K, Cd, H, W)).reshape(B, Cd, H, W)
    return layernorm2d_ref(y, sd[pre + "out_norm.weight"], sd[pre + "out_norm.bias"])


def ss2d_ref(sd: SD, pre: str, x, src=None, scan=selective_scan_ref):
    """SS2D.forwardv2 (vmamba.py:700-716) with disable_z (v05_noz)."""
    w, b = _wb(sd, pre + "in_proj.", src)
    x = F.conv2d(x, _w4(w), b)
    w, b = _wb(sd, pre + "conv2d.", src)
    x = F.silu(F.conv2d(x, w, b, padding=1, groups=x.shape[1]))
    y = ss2d_core_ref(sd, pre, x, scan)
    w, b = _wb(sd, pre + "out_proj.", src)
    return F.conv2d(y, _w4(w), b)


def gdmlp_ref(sd: SD, pre: str, x, src=None):
    """gdMlp (vmamba.py:116-133)."""
    w, b = _wb(sd, pre + "project_in.", src)
    x = F.conv2d(x, w, b)
    w, b = _wb(sd, pre + "dwconv.", src)
    x1, x2 = F.conv2d(x, w, b, padding=1, groups=x.shape[1]).chunk(2, dim=1)
    w, b = _wb(sd, pre + "project_out.", src)
    return F.conv2d(F.gelu(x1) * x2, w, b)


def vssblock_ref(sd: SD, pre: str, x, src=None, scan=selective_scan_ref):
    """VSSBlock._forwardv01 (vmamba.py:1319-1334), post_norm=False, drop_path=0."""
    x = x + ss2d_ref(sd, pre + "op.", layernorm2d_ref(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"]), src, scan)
    x = x + gdmlp_ref(sd, pre + "mlp.", layernorm2d_ref(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]), src)
    return x


def _blocks(sd: SD, pre: str, x, src, scan):
    i = 0
    while f"{pre}{i}.norm.weight" in sd:
        x = vssblock_ref(sd, f"{pre}{i}.", x, src, scan)
        i += 1
    assert i > 0, f"no VSSBlocks under {pre}"
    return x


# ----------------------------------------------------------------------------------------------
# A5  quaternion-Retinex decomposition   basicsr/QD/model4.py:167-262, model1.py,
#     wavelet-domain variant: basicsr/archs/DecompDualBranchDDWavelet_arch.py:80-132
# ----------------------------------------------------------------------------------------------

def quaternion_stack_ref(img):
    """RGB -> (q1_r,q2_r,q1_i,q2_i,q1_j,q2_j,q1_k,q2_k); q1 = RGB/(max_c+1e-7), q2 = RGB."""
    mx = img.max(dim=1, keepdim=True)[0]
    q1 = img / (mx + 1e-7)
    z = torch.zeros_like(mx)
    return torch.cat([z, z, q1[:, 0:1], img[:, 0:1], q1[:, 1:2], img[:, 1:2], q1[:, 2:3], img[:, 2:3]], dim=1)


def cross_attention_ref(sd: SD, pre: str, f1, f2):
    """SymmetricCrossAttention (model4.py:81-139), heads=1: channel attention, C x C logits over HW."""
    B, C, H, W = f1.shape
    pj = lambda n, t: F.conv2d(t, sd[pre + n + ".weight"], sd[pre + n + ".bias"]).reshape(B, C, H * W)
    scale = C ** -0.5
    q1, k2, v2 = pj("q1_proj", f1) * scale, pj("k2_proj", f2), pj("v2_proj", f2)
    q2, k1, v1 = pj("q2_proj", f2) * scale, pj("k1_proj", f1), pj("v1_proj", f1)
    c1 = (torch.softmax(q1 @ k2.transpose(1, 2), dim=-1) @ v2).reshape(B, C, H, W)
    c2 = (torch.softmax(q2 @ k1.transpose(1, 2), dim=-1) @ v1).reshape(B, C, H, W)
    o1 = F.conv2d(c1, sd[pre + "out1.weight"], sd[pre + "out1.bias"]) + f1
    o2 = F.conv2d(c2, sd[pre + "out2.weight"], sd[pre + "out2.bias"]) + f2
    return o1, o2


def decomp_trunk_ref(sd: SD, pre: str, img, model=None, mini_unet=True):
    """Shared trunk: quaternion stack -> DWT -> convs -> cross attention -> fuse -> conv_out +
    sharpening.  Returns the 32-channel wavelet-domain map (before IWT / index split).
    model: None / 'model1' / 'model4' (QD/model4.py:234-256) | 'model2' (second branch convs dilated by 2, QD/model2.py:171-181) |
    'model3' (mini U-Net in front of the branches, QD/model3.py:245-258; recognised by its down_conv weights too).  Dropout layers
    of model3 are identities in eval mode."""
    c3 = lambda n, t, **kw: F.conv2d(t, sd[pre + n + ".weight"], sd[pre + n + ".bias"], **({"padding": 1} | kw))
    feat = c3("conv_in", dwt_ref(quaternion_stack_ref(img)))
    if mini_unet and (model == "model3" or (model is None and pre + "down_conv.weight" in sd)):
        mid = F.relu(c3("mid_conv", F.relu(c3("down_conv", feat, stride=2))))
        feat = feat + F.conv_transpose2d(mid, sd[pre + "up_conv.weight"], sd[pre + "up_conv.bias"], stride=2)
    dil = dict(padding=2, dilation=2) if model == "model2" else {}
    f1 = c3("branch_q1.2", F.relu(c3("branch_q1.0", feat)), **dil) + feat
    f2 = c3("branch_q2.2", F.relu(c3("branch_q2.0", feat)), **dil) + feat
    f1, f2 = cross_attention_ref(sd, pre + "cross_attn.", f1, f2)
    fused = F.conv2d(torch.cat([f1, f2], 1), sd[pre + "fuse.weight"], sd[pre + "fuse.bias"])
    out = c3("conv_out", fused)
    return out + c3("sharpening", out)


def decomp_wavelet_ref(sd: SD, pre: str, img, model=None):
    """MyDecomp.forward (DecompDualBranchDDWavelet_arch.py:80-132): even channels -> Q1_w, odd -> Q2_w.  MyDecomp overrides the base
    class's forward with the model1 / model4 op sequence: for model3 the mini U-Net weights are loaded but never applied (:104-107)."""
    out = decomp_trunk_ref(sd, pre, img, model, mini_unet=False)
    return out[:, 0::2], out[:, 1::2]


def decomp_full_ref(sd: SD, pre: str, img, model=None):
    """Decomp.forward of model1 / model2 / model3 (no smoothing) / model4 (PostSmooth when smooth_q* weights exist)."""
    out = iwt_ref(decomp_trunk_ref(sd, pre, img, model))
    q1, q2 = out[:, [0, 2, 4, 6]], out[:, [1, 3, 5, 7]]
    if pre + "smooth_q1.conv.weight" in sd:
        sm = lambda n, t: t + F.relu(F.conv2d(t, sd[pre + n + ".conv.weight"], sd[pre + n + ".conv.bias"], padding=1, groups=4))
        q1, q2 = sm("smooth_q1", q1), sm("smooth_q2", q2)
    return q1, q2


# ----------------------------------------------------------------------------------------------
# A6  Stage-II networks
# ----------------------------------------------------------------------------------------------

def _levels(sd: SD, pre: str) -> int:
    n = 0
    while f"{pre}{n}.weight" in sd:
        n += 1
    return n


def ddwavelet_ref(sd: SD, x, scan=selective_scan_ref, pre: str = "", decomp_model=None):
    """DecompDualBranchDDWavelet.forward (DecompDualBranchDDWavelet_arch.py:301-369). Returns final_out."""
    img, cond = x[:, 0:3], x[:, 3:6]
    q1i, q2i = decomp_wavelet_ref(sd, pre + "decomp.", img, decomp_model)
    q1c, q2c = decomp_wavelet_ref(sd, pre + "decomp.", cond, decomp_model)
    nl = _levels(sd, pre + "down_layers_Q1.") + 1
    feats, skips = {}, {}
    for br, q in (("Q1", torch.cat([q1i, q1c], 1)), ("Q2", torch.cat([q2i, q2c], 1))):
        f = F.conv2d(q, sd[f"{pre}first_conv_{br}.weight"], sd[f"{pre}first_conv_{br}.bias"], padding=1)
        sk = []
        for i in range(nl - 1):
            f = _blocks(sd, f"{pre}encoders_{br}.{i}.", f, None, scan)
            sk.append(f)
            f = F.conv2d(f, sd[f"{pre}down_layers_{br}.{i}.weight"], None, stride=2, padding=1)
        feats[br], skips[br] = f, sk
    fused = F.conv2d(torch.cat([feats["Q1"], feats["Q2"]], 1), sd[pre + "bottleneck_fuse.weight"])
    fused = _blocks(sd, pre + "bottleneck_block.", fused, None, scan)
    outs = []
    for br in ("Q1", "Q2"):
        f = F.conv2d(fused, sd[f"{pre}bottleneck_to_{br}.weight"])
        for j in range(nl - 1):
            d = f"{pre}decoders_{br}.{j}."
            f = F.conv_transpose2d(f, sd[d + "up.weight"], sd[d + "up.bias"], stride=2)
            f = F.conv2d(torch.cat([f, skips[br][nl - 2 - j]], 1), sd[d + "fuse.weight"])
            f = _blocks(sd, d + "block.", f, None, scan)
        outs.append(iwt_ref(F.conv2d(f, sd[f"{pre}proj_{br}.weight"], sd[f"{pre}proj_{br}.bias"], padding=1)))
    return hamilton_ref(outs[0], outs[1])[:, 1:]


def singlebranch_ref(sd: SD, x, scan=selective_scan_ref, pre: str = "", decomp_model=None):
    """DecompSingleBranch.forward (DecompSingleBranch_arch.py:195-237). Returns final_out."""
    img, cond = x[:, :3], x[:, 3:]
    q1, q2 = decomp_full_ref(sd, pre + "decomp.", img, decomp_model)
    f = F.conv2d(torch.cat([q1, q2, cond], 1), sd[pre + "first_conv.weight"], sd[pre + "first_conv.bias"], padding=1)
    nl = _levels(sd, pre + "down_layers.") + 1
    sk = []
    for i in range(nl - 1):
        f = _blocks(sd, f"{pre}encoders.{i}.", f, None, scan)
        sk.append(f)
        f = F.conv2d(f, sd[f"{pre}down_layers.{i}.weight"], None, stride=2, padding=1)
    f = _blocks(sd, pre + "bottleneck.", f, None, scan)
    for j in range(nl - 1):
        d = f"{pre}decoders.{j}."
        f = F.conv_transpose2d(f, sd[d + "up.weight"], sd[d + "up.bias"], stride=2)
        f = F.conv2d(torch.cat([f, sk[nl - 2 - j]], 1), sd[d + "fuse.weight"])
        f = _blocks(sd, d + "block.", f, None, scan)
    out = F.conv2d(f, sd[pre + "proj.weight"], sd[pre + "proj.bias"], padding=1)
    return hamilton_ref(out[:, :4], out[:, 4:])[:, 1:]


def _dual_unet_ref(sd: SD, pre: str, q1, q2, scan):
    """Two-branch U-Net body shared by the dual-branch archs (…DD_arch.py:253-296): returns (Q1_out, Q2_out)."""
    nl = _levels(sd, pre + "down_layers_Q1.") + 1
    feats, skips = {}, {}
    for br, q in (("Q1", q1), ("Q2", q2)):
        f = F.conv2d(q, sd[f"{pre}first_conv_{br}.weight"], sd[f"{pre}first_conv_{br}.bias"], padding=1)
        sk = []
        for i in range(nl - 1):
            f = _blocks(sd, f"{pre}encoders_{br}.{i}.", f, None, scan)
            sk.append(f)
            f = F.conv2d(f, sd[f"{pre}down_layers_{br}.{i}.weight"], None, stride=2, padding=1)
        feats[br], skips[br] = f, sk
    fused = F.conv2d(torch.cat([feats["Q1"], feats["Q2"]], 1), sd[pre + "bottleneck_fuse.weight"])
    fused = _blocks(sd, pre + "bottleneck_block.", fused, None, scan)
    outs = []
    for br in ("Q1", "Q2"):
        f = F.conv2d(fused, sd[f"{pre}bottleneck_to_{br}.weight"])
        for j in range(nl - 1):
            d = f"{pre}decoders_{br}.{j}."
            f = F.conv_transpose2d(f, sd[d + "up.weight"], sd[d + "up.bias"], stride=2)
            f = F.conv2d(torch.cat([f, skips[br][nl - 2 - j]], 1), sd[d + "fuse.weight"])
            f = _blocks(sd, d + "block.", f, None, scan)
        outs.append(F.conv2d(f, sd[f"{pre}proj_{br}.weight"], sd[f"{pre}proj_{br}.bias"], padding=1))
    return outs


def dualbranch2dd_ref(sd: SD, x, scan=selective_scan_ref, pre: str = ""):
    """DecompDualBranch2DD.forward (DecompDualBranchDD_arch.py:239-299)."""
    q1i, q2i = decomp_full_ref(sd, pre + "decomp.", x[:, 0:3])
    q1c, q2c = decomp_full_ref(sd, pre + "decomp.", x[:, 3:6])
    o1, o2 = _dual_unet_ref(sd, pre, torch.cat([q1i, q1c], 1), torch.cat([q2i, q2c], 1), scan)
    return hamilton_ref(o1, o2)[:, 1:]


def dualbranch2_ref(sd: SD, x, scan=selective_scan_ref, pre: str = ""):
    """DecompDualBranch2.forward (DecompDualBranch_arch.py:231-298): Q = Q_img + [cond, 0]."""
    q1i, q2i = decomp_full_ref(sd, pre + "decomp.", x[:, 0:3])
    cq = torch.cat([x[:, 3:6], torch.zeros_like(x[:, 0:1])], 1)
    o1, o2 = _dual_unet_ref(sd, pre, q1i + cq, q2i + cq, scan)
    return hamilton_ref(o1, o2)[:, 1:]


def se_block_ref(sd: SD, pre: str, x):
    """SEBlock (DecompModel_arch.py:68-83): x * sigmoid(W2 relu(W1 mean_hw(x)))."""
    y = torch.relu(x.mean((2, 3)) @ sd[pre + "fc.0.weight"].t()) @ sd[pre + "fc.2.weight"].t()
    return x * torch.sigmoid(y)[:, :, None, None]


def spatial_attention_ref(sd: SD, pre: str, x):
    """SpatialAttention (DecompModel_arch.py:85-99): x * sigmoid(conv7x7([mean_c x, max_c x]))."""
    w = sd[pre + "conv.weight"]
    a = F.conv2d(torch.cat([x.mean(1, keepdim=True), x.max(1, keepdim=True)[0]], 1), w, padding=w.shape[-1] // 2)
    return x * torch.sigmoid(a)


def cross_fusion_ref(sd: SD, pre: str, x_src, x_tgt):
    """CrossFusionBlock (DecompModel_arch.py:57-66): x_tgt + gate * (W x_src + b)."""
    return x_tgt + sd[pre + "gate"] * F.conv2d(x_src, sd[pre + "transform.weight"], sd[pre + "transform.bias"])


def dualbranch_ref(sd: SD, x, scan=selective_scan_ref, pre: str = ""):
    """DecompDualBranch.forward (DecompModel_arch.py:292-352): the image's two quaternion maps (the condition channels are not read) through
    two U-Nets that meet once, by cross-fusion at the deepest encoder level (branch 2 first, branch 1 from the fused branch 2); SE + spatial
    attention after each bottleneck; Hamilton product of the two 4-channel outputs, imaginary parts."""
    q = decomp_full_ref(sd, pre + "decomp.", x[:, 0:3])
    nl = _levels(sd, pre + "down_layers.") + 1
    feats, skips = [], []
    for s_, qb in zip(("", "2"), q):
        f = F.conv2d(qb, sd[f"{pre}first_conv{s_}.weight"], sd[f"{pre}first_conv{s_}.bias"], padding=1)
        sk = []
        for i in range(nl - 1):
            f = _blocks(sd, f"{pre}encoders{s_}.{i}.", f, None, scan)
            sk.append(f)
            f = F.conv2d(f, sd[f"{pre}down_layers{s_}.{i}.weight"], None, stride=2, padding=1)
        feats.append(f); skips.append(sk)
    f2 = cross_fusion_ref(sd, pre + "cross_fusion_12.", feats[0], feats[1])
    f1 = cross_fusion_ref(sd, pre + "cross_fusion_21.", f2, feats[0])
    outs = []
    for s_, f, sk in (("", f1, skips[0]), ("2", f2, skips[1])):
        f = _blocks(sd, f"{pre}bottleneck{s_}.", f, None, scan)
        f = spatial_attention_ref(sd, f"{pre}spatial_attention{s_}.", se_block_ref(sd, f"{pre}bottleneck_se{s_}.", f))
        for j in range(nl - 1):
            d = f"{pre}decoders{s_}.{j}."
            f = F.conv_transpose2d(f, sd[d + "up.weight"], sd[d + "up.bias"], stride=2)
            f = F.conv2d(torch.cat([f, sk[nl - 2 - j]], 1), sd[d + "fuse.weight"])
            f = _blocks(sd, d + "block.", f, None, scan)
        outs.append(F.conv2d(f, sd[f"{pre}proj{s_}.weight"], sd[f"{pre}proj{s_}.bias"], padding=1))
    return hamilton_ref(outs[0], outs[1])[:, 1:]


def singlebranchdd_ref(sd: SD, x, scan=selective_scan_ref, pre: str = ""):
    """DecompSingleBranchDD.forward (DecompSingleBranchDD_arch.py:205-250)."""
    q1i, q2i = decomp_full_ref(sd, pre + "decomp.", x[:, 0:3])
    q1c, q2c = decomp_full_ref(sd, pre + "decomp.", x[:, 3:6])
    f = F.conv2d(torch.cat([q1i, q2i, q1c, q2c], 1), sd[pre + "first_conv.weight"], sd[pre + "first_conv.bias"], padding=1)
    nl = _levels(sd, pre + "down_layers.") + 1
    sk = []
    for i in range(nl - 1):
        f = _blocks(sd, f"{pre}encoders.{i}.", f, None, scan)
        sk.append(f)
        f = F.conv2d(f, sd[f"{pre}down_layers.{i}.weight"], None, stride=2, padding=1)
    f = _blocks(sd, pre + "bottleneck.", f, None, scan)
    for j in range(nl - 1):
        d = f"{pre}decoders.{j}."
        f = F.conv_transpose2d(f, sd[d + "up.weight"], sd[d + "up.bias"], stride=2)
        f = F.conv2d(torch.cat([f, sk[nl - 2 - j]], 1), sd[d + "fuse.weight"])
        f = _blocks(sd, d + "block.", f, None, scan)
    out = F.conv2d(f, sd[pre + "proj.weight"], sd[pre + "proj.bias"], padding=1)
    return hamilton_ref(out[:, :4], out[:, 4:])[:, 1:]


# ----------------------------------------------------------------------------------------------
# A1/A2  Stage-I Bayesian U-Net     basicsr/archs/UNet_arch.py:58-82,97-155,245-361,364-474
# ----------------------------------------------------------------------------------------------

def patch_merging_ref(sd: SD, pre: str, x):
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], 1)
    return F.conv2d(layernorm2d_ref(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"]), sd[pre + "reduction.weight"])


def dual_upsample_ref(sd: SD, pre: str, x):
    """DualUpSample(scale 2): pixel-shuffle branch || bilinear branch -> cat -> 1x1."""
    p = F.conv2d(x, sd[pre + "up_p.0.weight"])
    p = F.pixel_shuffle(F.prelu(p, sd[pre + "up_p.1.weight"]), 2)
    p = F.conv2d(p, sd[pre + "up_p.3.weight"])
    b = F.conv2d(x, sd[pre + "up_b.0.weight"], sd[pre + "up_b.0.bias"])
    b = F.interpolate(F.prelu(b, sd[pre + "up_b.1.weight"]), scale_factor=2, mode="bilinear", align_corners=False)
    b = F.conv2d(b, sd[pre + "up_b.3.weight"])
    return F.conv2d(torch.cat([p, b], 1), sd[pre + "conv.weight"])


def network_ref(sd: SD, x, src: Optional[EpsSource] = None, scan=selective_scan_ref, pre: str = "", mask=None):
    """Network.forward, one SubNetwork (stage=1), use_pixelshuffle=True.  src=None -> deterministic (mu) prediction; otherwise
    weights are sampled layer by layer in execution order, weight then bias (conv.py:106-110).  ``mask`` (B,H,W) is the
    masked-image-modelling mix of training mode (UNet_arch.py:463-466); eval mode never passes one."""
    fea0 = F.conv2d(x, sd[pre + "first_conv.weight"], sd[pre + "first_conv.bias"], padding=1)
    if mask is not None:
        w = mask.unsqueeze(1).type_as(fea0)
        fea0 = fea0 * (1.0 - w) + sd[pre + "mask_token"].expand(fea0.shape[0], -1, fea0.shape[2], fea0.shape[3]) * w
    s = pre + "subnets.0."
    nl = 0
    while f"{s}encoder_layers.{nl}.0.blocks.0.norm.weight" in sd:
        nl += 1
    f, enc = fea0, []
    for i in range(nl):
        f = _blocks(sd, f"{s}encoder_layers.{i}.0.blocks.", f, src, scan)
        enc.append(f)
        f = patch_merging_ref(sd, f"{s}encoder_layers.{i}.1.", f)
    f = _blocks(sd, s + "bottleneck.blocks.", f, src, scan)
    for j in range(nl):
        d = f"{s}decoder_layers.{j}."
        f = dual_upsample_ref(sd, d + "0.", f)
        f = F.conv2d(torch.cat([f, enc[nl - 1 - j]], 1), sd[d + "1.weight"])
        f = _blocks(sd, d + "2.blocks.", f, src, scan)
    f = fea0 + f
    return F.conv2d(f, sd[pre + "proj.weight"], sd[pre + "proj.bias"], padding=1)


# ----------------------------------------------------------------------------------------------
# A0/A3/A4/A12  eval driver pieces         Enhancement/eval.py:146-153,170-176,199-297, utils.py:5-9
# ----------------------------------------------------------------------------------------------

def pad_reflect_ref(img_hwc, factor):
    """_padimg_np (eval.py:146-153): reflect-pad bottom/right to the next multiple of ``factor``."""
    import numpy as np
    h, w = img_hwc.shape[:2]
    ph = (((h + factor) // factor) * factor - h) if h % factor else 0
    pw = (((w + factor) // factor) * factor - w) if w % factor else 0
    return np.pad(img_hwc, ((0, ph), (0, pw), (0, 0)), "reflect") if (ph or pw) else img_hwc


def cv2_resize_down(x, s):
    """cv2.resize(img, None, fx=1/s, fy=1/s, INTER_LINEAR) for integer s on an NCHW tensor whose
    H, W are multiples of s (eval.py:174).  With source coordinate (i+0.5)*s-0.5 the bilinear
    taps are pixels s*i + s/2 - 1 and s*i + s/2 with weight 0.5 each (even s).  PARITY UNPINNED:
    cv2 is not installed here; this is the definition of INTER_LINEAR without anti-aliasing."""
    assert s % 2 == 0
    a = s // 2 - 1
    r = 0.5 * (x[:, :, a::s, :] + x[:, :, a + 1::s, :])
    return 0.5 * (r[:, :, :, a::s] + r[:, :, :, a + 1::s])


def psnr_ref(target, pred):
    """Enhancement/utils.py:5-9 on float [0,1] arrays."""
    import numpy as np
    mse = float(np.mean((np.asarray(target) - np.asarray(pred)) ** 2))
    return 100.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)


def eval_mc_ref(sd1: SD, sd2: SD, img, target, num_samples, *, scale=16, noise_level=0.1,
                gt_mean=True, deterministic=False, eps_list=None, noise_list=None,
                stage2=ddwavelet_ref, scan=selective_scan_ref, img_down=None, generator=None):
    """The per-image Monte-Carlo loop of Enhancement/eval.py:170-297 (condition type 'mean',
    full-reference selection with psnr_weight=1).

    img, target: (1,3,h,w) float [0,1] tensors (target may be None when gt_mean=False).
    eps_list[i]: recorded epsilon dict for Stage-I sample i (or None -> deterministic);
    noise_list[i]: the (1,3,hp/scale,wp/scale) N(0,1) draw added to sample i's condition.
    Returns dict(conds, preds (pre GT-mean, clamped), finals, psnr list, best index)."""
    import numpy as np
    _, _, h, w = img.shape
    pad = torch.from_numpy(np.ascontiguousarray(
        pad_reflect_ref(img[0].permute(1, 2, 0).numpy(), 4 * scale))).permute(2, 0, 1)[None]
    if img_down is None:
        img_down = cv2_resize_down(pad, scale)
    if deterministic:
        num_samples = 1
    conds = []
    for i in range(num_samples):
        src = None if deterministic else EpsSource(None if eps_list is None else eps_list[i], generator)
        c = network_ref(sd1, img_down, src, scan).clamp(0, 1)
        if gt_mean:
            c = (c * (target.mean(dim=(2, 3), keepdim=True) / c.mean(dim=(2, 3), keepdim=True))).clamp(0, 1)
        nz = noise_list[i] if noise_list is not None else torch.randn(c.shape, generator=generator)
        conds.append(c + nz * noise_level)
    preds, finals, psnrs = [], [], []
    tgt = None if target is None else target[0].permute(1, 2, 0).numpy()
    for c in conds:
        up = F.interpolate(c, scale_factor=scale, mode="bilinear", align_corners=False)
        p = stage2(sd2, torch.cat([pad, up], 1), scan)[:, :, :h, :w].clamp(0, 1)
        preds.append(p)
        q = p[0].permute(1, 2, 0).numpy()
        if gt_mean:
            q = np.clip(q * (tgt.mean(axis=(0, 1), keepdims=True) / q.mean(axis=(0, 1), keepdims=True)), 0, 1)
        finals.append(q)
        if tgt is not None:
            psnrs.append(psnr_ref(tgt, q))
    best = 0
    if psnrs:
        rel = (np.array(psnrs) / max(psnrs)).tolist()      # eval.py:284 with psnr_weight=1
        best = rel.index(max(rel))
    return dict(img_down=img_down, conds=conds, preds=preds, finals=finals, psnr=psnrs, best=best)


# ----------------------------------------------------------------------------------------------
# A10  Stage-II training step      basicsr/models/image_enhancer_model.py:143-148,165-216
#      (L1 pixel loss only: the perceptual term needs VGG19 weights that cannot be fetched here)
# ----------------------------------------------------------------------------------------------

def train_step_ref(sd: SD, lq, gt, conds, *, stage2=ddwavelet_ref, scale=16, lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
                   weight_decay=1e-4, max_grad_norm=1.0, steps=1, scan=selective_scan_ref):
    """``optimize_parameters``: zero_grad -> up_conds = interpolate(conds, x scale) -> net_g(cat[lq, up_conds]) -> L1 ->
    backward -> clip_grad_norm_(max_grad_norm) -> AdamW.step (optim_g of Options/DecompDualBranch2DDWavelet_4.yml:88-92).
    ``sd``: the net's state dict; every key outside ``decomp.`` is trainable (the frozen decomposition is under no_grad,
    DecompDualBranchDDWavelet_arch.py:307).  Returns per step the loss, the total gradient norm before clipping, the unclipped
    gradients of the first step, and the parameters after the last step."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if not k.startswith("decomp.")}
    frozen = {k: v.detach() for k, v in sd.items() if k.startswith("decomp.")}
    opt = torch.optim.AdamW(list(params.values()), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    losses, norms, grads0 = [], [], None
    for s in range(steps):
        opt.zero_grad()
        up = F.interpolate(conds, scale_factor=scale, mode="bilinear", align_corners=False)
        out = stage2({**frozen, **params}, torch.cat([lq, up], 1), scan)
        loss = (out - gt).abs().mean()
        loss.backward()
        if s == 0:
            grads0 = {k: p.grad.detach().clone() for k, p in params.items()}
        if max_grad_norm:
            gn = torch.nn.utils.clip_grad_norm_(list(params.values()), max_grad_norm)
        else:
            gn = torch.sqrt(sum((p.grad ** 2).sum() for p in params.values()))
        opt.step()
        losses.append(float(loss.detach()))
        norms.append(float(gn))
    return dict(loss=losses, grad_norm=norms, grads=grads0, params={k: v.detach() for k, v in params.items()}, out=out.detach())


# ----------------------------------------------------------------------------------------------
# (f)2  Stage-I training step     basicsr/models/condition_generator_model.py:176-218,
#       basicsr/bayesian/conv.py:78-114, linear.py:55-90, base_layer.py:26-40, tools.py:76-84
# ----------------------------------------------------------------------------------------------

def kl_div_ref(mu_q, sigma_q, mu_p, sigma_p):
    return (torch.log(sigma_p) - torch.log(sigma_q) + (sigma_q ** 2 + (mu_q - mu_p) ** 2) / (2 * (sigma_p ** 2)) - 0.5).mean()


def bnn_layers_ref(sd: SD):
    """prefixes ('...in_proj.') of the Bayesian leaves in module order (= state-dict order)."""
    return [k[:-len("mu_weight")] for k in sd if k.endswith("mu_weight")]


def stage1_train_step_ref(sd: SD, prior: SD, lq, gt, eps_steps, masks, *, mini_batch=8, kl_weight=0.01, decay=0.998, lr=2e-4,
                          betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0, bnn_step0=0, scan=selective_scan_ref):
    """``ConditionGenerator.optimize_parameters`` for ``len(eps_steps)`` iterations.  ``sd``: the BNN-converted Network's state dict
    (every entry is a parameter); ``prior``: {'<layer>.prior_mu_weight' ...} buffers; ``eps_steps[i]``: the N(0,1) draws of iteration i
    ('<layer>.weight' / '.bias'); ``masks[i]``: MIM mask or None.  Per training forward every Bayesian leaf first moves its prior
    (threshold EMA with d = min(decay, (1 + t) / (10 + t)), t = forwards so far), then samples w = mu + softplus(rho) eps; the loss is
    kl_weight * sum_layers KL(q || prior) / mini_batch + L1.  A parameter that took no part in the graph keeps ``grad = None`` and is
    skipped by AdamW (mask_token in an iteration without a mask)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    prior = {k: v.detach().clone() for k, v in prior.items()}
    layers = bnn_layers_ref(sd)
    opt = torch.optim.AdamW(list(params.values()), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
    kls, pixs, norms, grads0, out = [], [], [], None, None
    for it, (e, m) in enumerate(zip(eps_steps, masks)):
        opt.zero_grad()
        t = bnn_step0 + it
        d = min(decay, (1 + t) / (10 + t))
        with torch.no_grad():
            for L in layers:
                for kind in ("weight", "bias"):
                    if L + "mu_" + kind in params:
                        for mr in ("mu", "rho"):
                            prior[f"{L}prior_{mr}_{kind}"] = d * prior[f"{L}prior_{mr}_{kind}"] + (1 - d) * params[f"{L}{mr}_{kind}"].detach()
        out = network_ref(params, lq, EpsSource(e), scan, mask=m)
        kl = None
        for L in layers:
            for kind in ("weight", "bias"):
                if L + "mu_" + kind in params:
                    term = kl_div_ref(params[f"{L}mu_{kind}"], torch.log1p(torch.exp(params[f"{L}rho_{kind}"])),
                                      prior[f"{L}prior_mu_{kind}"], torch.log1p(torch.exp(prior[f"{L}prior_rho_{kind}"])))
                    kl = term if kl is None else kl + term
        pix = (out - gt).abs().mean()
        (kl_weight * kl / mini_batch + pix).backward()
        if it == 0:
            grads0 = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in params.items()}
        live = [p for p in params.values() if p.grad is not None]
        norms.append(float(torch.nn.utils.clip_grad_norm_(live, max_grad_norm)) if max_grad_norm else
                     float(torch.sqrt(sum((p.grad ** 2).sum() for p in live))))
        opt.step()
        kls.append(float(kl.detach())); pixs.append(float(pix.detach()))
    return dict(l_kl=kls, l_pix=pixs, grad_norm=norms, grads=grads0, params={k: v.detach() for k, v in params.items()}, prior=prior,
                out=out.detach())


# ----------------------------------------------------------------------------------------------
# A12  selection metrics and rules      Enhancement/utils.py:12-57, Enhancement/eval.py:224-225,268-297,308-314
# ----------------------------------------------------------------------------------------------

def ssim_ref(img1, img2):
    """calculate_ssim(img1, img2) for (h,w,3) uint8-valued arrays: per channel, float64, cv2.getGaussianKernel(11, 1.5) outer product,
    cv2.filter2D(...)[5:-5, 5:-5] (the valid region: the border mode never matters), constants (0.01*255)^2, (0.03*255)^2.
    cv2 is not installed anywhere here; the Gaussian kernel is its documented closed form (normalised exp(-(i-5)^2 / (2 sigma^2)) for
    ksize 11 > 7), the correlation is restated with numpy -- values parity-unpinned against cv2 itself, formula from utils.py."""
    import numpy as np
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    win = np.outer(g, g)

    def filt(a):
        h, w = a.shape
        out = np.zeros((h - 10, w - 10))
        for i in range(11):
            for j in range(11):
                out += win[i, j] * a[i:i + h - 10, j:j + w - 10]
        return out
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for c in range(3):
        a, b = img1[:, :, c].astype(np.float64), img2[:, :, c].astype(np.float64)
        mu1, mu2 = filt(a), filt(b)
        s1, s2, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
        vals.append((((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean())
    return float(np.mean(vals))


def img_as_ubyte_ref(x):
    """skimage.util.img_as_ubyte for float images in [0,1]: rint(255 x) (round half to even) as uint8."""
    import numpy as np
    return np.rint(np.clip(x, 0, 1) * 255.0).astype(np.uint8)


def select_ref(psnr_list=None, ssim_list=None, psnr_weight=1.0, no_ref_list=None, no_ref="", ):
    """eval.py:268-297: index of the chosen candidate.  no_ref 'clip' -> index(max), 'niqe' -> index(min); otherwise the weighted
    full-reference rule :284-285.  Python list semantics (first occurrence)."""
    import numpy as np
    if no_ref == "clip":
        return no_ref_list.index(max(no_ref_list))
    if no_ref == "niqe":
        return no_ref_list.index(min(no_ref_list))
    best = (psnr_weight * np.array(psnr_list) / max(psnr_list) + (1 - psnr_weight) * np.array(ssim_list) / max(ssim_list)).tolist()
    return best.index(max(best))


def mc_mean_ref(preds_hw3, target_hw3=None, gt_mean=False):
    """eval.py:224-225,308-314: preds (N,h,w,3) clamped candidates -> clamp(mean) and the gray-mean GT rescale
    (cv2.COLOR_BGR2GRAY weights 0.114, 0.587, 0.299 on the stored channel order)."""
    import numpy as np
    mc = np.clip(np.mean(np.clip(preds_hw3, 0, 1), axis=0), 0, 1)
    if gt_mean:
        wts = np.array([0.114, 0.587, 0.299], dtype=np.float32)
        gm = float((mc.astype(np.float32) * wts).sum(-1).mean())
        gtm = float((target_hw3.astype(np.float32) * wts).sum(-1).mean())
        mc = np.clip(mc * (gtm / gm), 0, 1)
    return mc
