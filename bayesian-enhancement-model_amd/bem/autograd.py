"""Training path of the Stage-II nets: ``torch.autograd.Function``s whose forward AND backward are sequences of hand-written
gfx950 kernels (bem.ops).  torch's autograd engine only walks the graph; no gradient is computed by a torch op.

What is differentiated: ``DecompDualBranchDDWavelet.forward`` and its siblings
(basicsr/archs/DecompDualBranchDDWavelet_arch.py:301-369) as driven by ``ImageEnhancer.optimize_parameters``
(basicsr/models/image_enhancer_model.py:165-216): Hamilton product / IWT, dense 3x3 and 4x4-stride-2 convolutions,
ConvTranspose2d(2,2), 1x1 fuse layers and the 18 ``VSSBlock``s (vmamba.py:1319-1334: LayerNorm2d, Linear2d, depthwise 3x3 +
SiLU, the four-direction selective scan with x_proj / dt_proj -- the reference's CrossScanF / SelectiveScanCuda /
CrossMergeF backward, csm_triton.py:207-273, csms6s.py:95-113 -- and the gdMlp).

Parameter gradients are accumulated by the kernels straight into ``param.grad`` (allocated zero-filled on first use, or a view
of the flat buffer of ``bem.train.FlatParams``); the Functions return ``None`` for parameter inputs.  Use ``loss.backward()``
(as the reference's training step does) -- ``torch.autograd.grad`` with respect to parameters is not supported.
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function

from . import ops
from .native import BemNativeError
from .native import check, lib

WEIGHT_EPOCH = ops.WEIGHT_EPOCH      # see bem.ops: bumped by optimizer steps that rewrite parameters in place


class _Derived:
    """Cache of tensors derived from parameters, valid for one weight epoch and parameter version."""

    def __init__(self):
        self.d = {}

    def get(self, key, srcs, fn):
        sig = (WEIGHT_EPOCH[0],) + tuple((t.data_ptr(), ops.tensor_version(t)) for t in srcs)
        hit = self.d.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        with torch.no_grad():
            val = fn()
        self.d[key] = (sig, val)
        return val


def _derived(holder) -> _Derived:
    c = getattr(holder, "_bem_derived", None)
    if c is None:
        c = _Derived()
        object.__setattr__(holder, "_bem_derived", c)
    return c


def grad_of(p: torch.Tensor) -> torch.Tensor:
    """The buffer the kernels accumulate this parameter's gradient into."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    elif not p.grad.is_contiguous():
        raise RuntimeError("bem.autograd: parameter .grad must be contiguous")
    return p.grad


def _pack(holder, key, srcs, make):
    """pack_pw_weight of ``make()`` (an (M,K) matrix), cached on ``holder``."""
    return _derived(holder).get(key, srcs, lambda: ops.pack_pw_weight(make() if ops.USE_X6 else make().contiguous()))


def _w2d(w):
    return w.detach().reshape(w.shape[0], -1)


def wb(m):
    """(weight, bias | None) tensors of a leaf: nn.Conv2d / nn.Linear parameters, or -- for a Bayesian leaf in a training forward -- the
    weights it sampled for this forward (modules._BayesBase.train_sample); their .grad is the buffer the reparameterisation kernel
    later folds into mu / rho."""
    if hasattr(m, "mu_weight"):
        if m.deterministic:
            return m.mu_weight, (m.mu_bias if m.bias else None)
        if getattr(m, "_ws", None) is None:
            raise RuntimeError("Bayesian leaf: no weight sample is active (training forwards run under Network.forward)")
        return m._ws, m._bs
    return m.weight, m.bias


# ------------------------------------------------------------------------------------------------------------------
# small nodes
# ------------------------------------------------------------------------------------------------------------------
class ForkFn(Function):
    """A tensor with two consumers: the two incoming gradients are summed by bem_add_f32 (not by the engine's own add)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g1, g2):
        if g1 is None:
            return g2
        if g2 is None:
            return g1
        return ops.add(g1.contiguous(), g2.contiguous())


def fork(x):
    return ForkFn.apply(x) if x.requires_grad else (x, x)


class L1LossFn(Function):
    """basicsr/losses/losses.py L1Loss(loss_weight, reduction='mean')."""

    @staticmethod
    def forward(ctx, pred, gt, weight):
        pred, gt = pred.contiguous(), gt.contiguous()
        ctx.save_for_backward(pred, gt)
        ctx.weight = weight
        return ops.l1_loss(pred, gt, weight).reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, gt = ctx.saved_tensors
        # g = dL/dloss as a device scalar (1.0 in the reference's step; an AMP scaler passes its scale): folded into the kernel
        return ops.l1_loss_bwd(pred, gt, ctx.weight, g.reshape(1).contiguous().float()), None, None


def l1_loss(pred, gt, weight=1.0):
    return L1LossFn.apply(pred, gt, float(weight))


class IwtHamiltonFn(Function):
    @staticmethod
    def forward(ctx, q1w, q2w):
        q1w, q2w = q1w.contiguous(), q2w.contiguous()
        ctx.save_for_backward(q1w, q2w)
        return ops.iwt_hamilton(q1w, q2w)

    @staticmethod
    def backward(ctx, dout):
        q1w, q2w = ctx.saved_tensors
        return ops.iwt_hamilton_bwd(q1w, q2w, dout.contiguous())


# ------------------------------------------------------------------------------------------------------------------
# 1x1 layers without a LayerNorm prologue (fuse, bottleneck_fuse, bottleneck_to_Q*)
# ------------------------------------------------------------------------------------------------------------------
class PwFn(Function):
    @staticmethod
    def forward(ctx, x1, x2, weight, bias, holder):
        x1 = x1.contiguous()
        M = weight.shape[0]
        Wp, _ = holder.gemm_weights(x1.shape[0])
        if x2 is not None:
            x2 = x2.contiguous()
            out = ops.pw_gemm(x1, Wp, M, x2=x2, in_mode=2, bias=None if bias is None else bias.detach())
        else:
            out = ops.pw_gemm(x1, Wp, M, bias=None if bias is None else bias.detach())
        ctx.save_for_backward(x1, x2)
        ctx.holder, ctx.weight, ctx.bias = holder, weight, bias
        return out

    @staticmethod
    def backward(ctx, dout):
        x1, x2 = ctx.saved_tensors
        dout = dout.contiguous()
        w, b, h = ctx.weight, ctx.bias, ctx.holder
        C1 = x1.shape[1]
        ops.pw_wgrad_(dout, x1, grad_of(w), x2=x2, dbias=None if b is None else grad_of(b))
        w2 = _w2d(w)
        dx1 = dx2 = None
        if ctx.needs_input_grad[0]:
            dx1 = ops.pw_gemm(dout, _pack(h, "T1", [w], lambda: w2[:, :C1].t()), C1)
        if x2 is not None and ctx.needs_input_grad[1]:
            C2 = x2.shape[1]
            dx2 = ops.pw_gemm(dout, _pack(h, "T2", [w], lambda: w2[:, C1:].t()), C2)
        return dx1, dx2, None, None, None


class ConvT2x2Fn(Function):
    """nn.ConvTranspose2d(C, C/2, 2, 2): forward = 1x1 GEMM to 4*Co rows + 2x2 scatter (modules.ConvT2x2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, holder):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.holder, ctx.weight, ctx.bias = holder, weight, bias
        return holder._forward_nograd(x)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        w, b, h = ctx.weight, ctx.bias, ctx.holder
        Cin, Co = w.shape[0], w.shape[1]
        dy4 = ops.pixel_unshuffle2(dout.contiguous())                 # (B, 4 Co, H, W), channel = co*4 + dy*2 + dx
        ops.pw_wgrad_(x, dy4, grad_of(w))                            # (Cin, Co*4) = weight.grad's own layout
        ops.channel_sum_(dout, grad_of(b))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.pw_gemm(dy4, _pack(h, "T", [w], lambda: w.detach().reshape(Cin, 4 * Co)), Cin)
        return dx, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# dense convolutions (3x3 stride 1 pad 1; 4x4 stride 2 pad 1)
# ------------------------------------------------------------------------------------------------------------------
def _wflip3(w):
    return w.detach().flip(2, 3).transpose(0, 1).contiguous()          # (Cin, Cout, 3, 3): correlation with the flipped kernel


def _wT4(w):
    """Input gradient of the 4x4 stride-2 pad-1 convolution as a 3x3 convolution over dout producing 4*Cin phase channels
    (PixelShuffle order ci*4 + py*2 + px), then pixel_shuffle2."""
    w = w.detach()
    Co, Ci = w.shape[0], w.shape[1]
    w3 = torch.zeros(Ci, 2, 2, Co, 3, 3, device=w.device, dtype=w.dtype)
    taps = {0: ((0, 1), (-1, 3)), 1: ((1, 0), (0, 2))}                 # phase -> ((offset, kernel index), ...)
    for py in (0, 1):
        for di, ky in taps[py]:
            for px in (0, 1):
                for dj, kx in taps[px]:
                    w3[:, py, px, :, di + 1, dj + 1] = w[:, :, ky, kx].t()
    return w3.reshape(Ci * 4, Co, 3, 3).contiguous()


class Conv2dFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, holder, cin_slice):
        if holder.dilation[0] != 1:
            raise BemNativeError("Conv2dFn: the training path covers undilated convolutions only (the dilated QD model2 layers are frozen)")
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.holder, ctx.weight, ctx.bias, ctx.cin_slice = holder, weight, bias, cin_slice
        return ops.conv2d(x, weight.detach(), None if bias is None else bias.detach(), stride=holder.stride[0], pad=holder.padding[0],
                          cin_slice=cin_slice)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        w, b, h = ctx.weight, ctx.bias, ctx.holder
        dout = dout.contiguous()
        s, p = h.stride[0], h.padding[0]
        ops.conv_wgrad_(dout, x, grad_of(w), None if b is None else grad_of(b), stride=s, pad=p, cin_slice=ctx.cin_slice)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.cin_slice is not None:
                raise NotImplementedError("Conv2dFn: input gradient of a channel-sliced input")
            k = tuple(w.shape[2:])
            if (k, s, p) == ((3, 3), 1, 1):
                dx = ops.conv2d(dout, _derived(h).get("flip", [w], lambda: _wflip3(w)), None, stride=1, pad=1)
            elif (k, s, p) == ((4, 4), 2, 1):
                d4 = ops.conv2d(dout, _derived(h).get("T4", [w], lambda: _wT4(w)), None, stride=1, pad=1)
                dx = ops.pixel_shuffle2(d4)
            else:
                raise NotImplementedError(f"Conv2dFn: input gradient for kernel {k} stride {s} pad {p}")
        return dx, None, None, None, None


# ------------------------------------------------------------------------------------------------------------------
# VSSBlock (vmamba.py:1319-1334) as one node
# ------------------------------------------------------------------------------------------------------------------
def _transpose_into(src, dst, dst_c0):
    """dst[:, dst_c0 : dst_c0 + C] (B,*,H,W) = planes of src (B,C,W,H) transposed."""
    B, C, Ws, Hs = src.shape
    L = Ws * Hs
    check(lib().bem_transpose_planes_f32(ctypes.c_void_p(src.data_ptr()), C * L, ctypes.c_void_p(dst.data_ptr() + 4 * dst_c0 * L),
                                         dst.shape[1] * L, B, C, Ws, Hs, ops._stream()), "transpose_planes(into)")


class VSSBlockFn(Function):
    @staticmethod
    def forward(ctx, x, blk, *params):
        from .modules import _out_features
        x = x.contiguous()
        op, mlp = blk.op, blk.mlp
        B, C, H, W = x.shape
        Ci, R, L = op.d_inner, op.dt_rank, H * W
        n1, n2, on = blk.norm, blk.norm2, op.out_norm
        Wp, b = op.in_proj.gemm_weights(B)
        t = ops.pw_gemm(x, Wp, Ci, ln=(n1.weight.detach(), n1.bias.detach()), ln_eps=n1.eps, bias=b)
        cw, cb = op.conv2d.dw_weights(B)
        xc = ops.dwconv3x3(t, cw, cb, mode=1)
        wall, dtw, dtb, A, Ds = op._scan_params()
        xd = ops.pw_gemm(xc, wall, 4 * (R + 2))
        xd1 = ops.transpose_plane_slice(xd, 2 * (R + 2), 2 * (R + 2))
        xcT = None
        if ops.ss2d_scan_rm_supported(H, W, R):
            y0, y1 = ops.ss2d_scan_rm(xc, xd.view(B, 4, R + 2, L)[:, :2], xd1.view(B, 2, R + 2, L), dtw, dtb, A, Ds)
        else:
            xcT = ops.transpose_planes(xc)
            y0, y1T = ops.ss2d_scan(xc.view(B, Ci, L), xcT.view(B, Ci, L), xd.view(B, 4, R + 2, L)[:, :2], xd1.view(B, 2, R + 2, L), dtw, dtb, A, Ds)
            y0, y1 = y0.view(B, Ci, H, W), ops.transpose_planes(y1T.view(B, Ci, W, H))
        Wp, b = op.out_proj.gemm_weights(B)
        x2 = ops.pw_gemm(y0, Wp, _out_features(op.out_proj), x2=y1, in_mode=1, ln=(on.weight.detach(), on.bias.detach()), ln_eps=on.eps, bias=b, res=x)
        Wp, b = mlp.project_in.gemm_weights(B)
        t2 = ops.pw_gemm(x2, Wp, _out_features(mlp.project_in), ln=(n2.weight.detach(), n2.bias.detach()), ln_eps=n2.eps, bias=b)
        dww, dwb = mlp.dwconv.dw_weights(B)
        g = ops.dwconv3x3(t2, dww, dwb, mode=2)
        Wp, b = mlp.project_out.gemm_weights(B)
        out = ops.pw_gemm(g, Wp, _out_features(mlp.project_out), bias=b, res=x2)
        ctx.blk = blk
        ctx.save_for_backward(x, t, xc, xd, xd1, y0, y1, x2, t2, g)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        x, t, xc, xd, xd1, y0, y1, x2, t2, g = ctx.saved_tensors
        dout = dout.contiguous()
        op, mlp = blk.op, blk.mlp
        B, C, H, W = x.shape
        Ci, R, L = op.d_inner, op.dt_rank, H * W
        n1, n2, on = blk.norm, blk.norm2, op.out_norm
        pi, dwc, po = mlp.project_in, mlp.dwconv, mlp.project_out
        (piw, pib), (dww, dwb), (pow_, pob) = wb(pi), wb(dwc), wb(po)
        Hd = pow_.shape[1]
        # ---- gdMlp: out = x2 + W_o g + b_o, g = GELU(h1) h2, h = dw(t2) + b, t2 = W_i LN2(x2) + b_i
        ops.pw_wgrad_(dout, g, grad_of(pow_), dbias=None if pob is None else grad_of(pob))
        dg = ops.pw_gemm(dout, _pack(po, "T", [pow_], lambda: _w2d(pow_).t()), Hd)
        dh = ops.dwact_bwd(t2, dww.detach(), None if dwb is None else dwb.detach(), dg, grad_of(dww), None if dwb is None else grad_of(dwb), 2)
        del dg
        dt2 = ops.dwconv3x3(dh, _derived(dwc).get("flip", [dww], lambda: dww.detach().flip(2, 3).contiguous()), None, mode=0)
        del dh
        dn2 = ops.pw_gemm(dt2, _pack(pi, "T", [piw], lambda: _w2d(piw).t()), C)
        dx2, nrm = ops.ln_bwd(x2, dn2, n2.weight.detach(), n2.bias.detach(), n2.eps, grad_of(n2.weight), grad_of(n2.bias), dres=dout)
        del dn2
        ops.pw_wgrad_(dt2, nrm, grad_of(piw), dbias=None if pib is None else grad_of(pib))
        del dt2, nrm
        # ---- SS2D: x2 = x + W_out LN_on(y0 + y1)
        opw = op.out_proj
        oww, owb = wb(opw)
        dysn = ops.pw_gemm(dx2, _pack(opw, "T", [oww], lambda: _w2d(oww).t()), Ci)
        dys, nys = ops.ln_bwd(y0, dysn, on.weight.detach(), on.bias.detach(), on.eps, grad_of(on.weight), grad_of(on.bias), x2=y1)
        del dysn
        ops.pw_wgrad_(dx2, nys, grad_of(oww), dbias=None if owb is None else grad_of(owb))
        del nys
        wall, dtw, dtb, A, Ds = op._scan_params()
        dysT = ops.transpose_planes(dys)
        xcT = ops.transpose_planes(xc)
        dx0, dx1T, dxd0, dxd1 = ops.ss2d_scan_bwd(
            xc.view(B, Ci, L), xcT.view(B, Ci, L), xd.view(B, 4, R + 2, L)[:, :2], xd1.view(B, 2, R + 2, L), dys.view(B, Ci, L), dysT.view(B, Ci, L),
            dtw, dtb, A, Ds, grad_of(op.A_logs), grad_of(op.Ds), grad_of(op.dt_projs_weight), grad_of(op.dt_projs_bias))
        del dysT, xcT, dys
        # x_dbl gradient back in the stacked row order of the forward GEMM: [dir 0 | dir 2 | dir 1 | dir 3], row-major pixels
        dxd = torch.empty(B, 4 * (R + 2), H, W, device=x.device, dtype=x.dtype)
        ops.copy_channels(dxd0.view(B, 2 * (R + 2), H, W), dxd, 0)
        _transpose_into(dxd1.view(B, 2 * (R + 2), W, H), dxd, 2 * (R + 2))
        ops.pw_wgrad_(dxd, xc, grad_of(op.x_proj_weight), blk_rows=R + 2, perm=(0, 2, 1, 3))
        xw = op.x_proj_weight
        wallT = _pack(op, "wallT", [xw], lambda: torch.cat([xw.detach()[0], xw.detach()[2], xw.detach()[1], xw.detach()[3]], 0).t())
        dxc = ops.pw_gemm(dxd, wallT, Ci, res=dx0.view(B, Ci, H, W))
        dxc = ops.add(dxc, ops.transpose_planes(dx1T.view(B, Ci, W, H)))
        del dx0, dx1T, dxd, dxd0, dxd1
        cv = op.conv2d
        cvw, cvb = wb(cv)
        dpre = ops.dwact_bwd(t, cvw.detach(), None if cvb is None else cvb.detach(), dxc, grad_of(cvw), None if cvb is None else grad_of(cvb), 1)
        dt = ops.dwconv3x3(dpre, _derived(cv).get("flip", [cvw], lambda: cvw.detach().flip(2, 3).contiguous()), None, mode=0)
        del dpre, dxc
        ipw = op.in_proj
        iww, iwb = wb(ipw)
        dn = ops.pw_gemm(dt, _pack(ipw, "T", [iww], lambda: _w2d(iww).t()), C)
        need_dx = ctx.needs_input_grad[0]
        dx, nrm = ops.ln_bwd(x, dn, n1.weight.detach(), n1.bias.detach(), n1.eps, grad_of(n1.weight), grad_of(n1.bias), dres=dx2)
        ops.pw_wgrad_(dt, nrm, grad_of(iww), dbias=None if iwb is None else grad_of(iwb))
        return (dx if need_dx else None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def vssblock_params(blk):
    return [p for p in blk.parameters() if p.requires_grad]


def vssblock(blk, x):
    return VSSBlockFn.apply(x, blk, *vssblock_params(blk))


# ------------------------------------------------------------------------------------------------------------------
# Stage-I U-Net pieces (basicsr/archs/UNet_arch.py): PatchMerging, DualUpSample, MIM token mix, outer residual
# ------------------------------------------------------------------------------------------------------------------
class AddFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class LnPwFn(Function):
    """LayerNorm2d + bias-free / biased 1x1 layer (PatchMerging.reduction(norm(x)), UNet_arch.py:80)."""

    @staticmethod
    def forward(ctx, x, norm, layer, *params):
        x = x.contiguous()
        Wp, b = layer.gemm_weights(x.shape[0])
        ctx.save_for_backward(x)
        ctx.norm, ctx.layer = norm, layer
        return ops.pw_gemm(x, Wp, layer.weight.shape[0], ln=(norm.weight.detach(), norm.bias.detach()), ln_eps=norm.eps, bias=b)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        n, m = ctx.norm, ctx.layer
        dout = dout.contiguous()
        w, b = wb(m)
        dn = ops.pw_gemm(dout, _pack(m, "T", [w], lambda: _w2d(w).t()), x.shape[1])
        dx, nrm = ops.ln_bwd(x, dn, n.weight.detach(), n.bias.detach(), n.eps, grad_of(n.weight), grad_of(n.bias))
        ops.pw_wgrad_(dout, nrm, grad_of(w), dbias=None if b is None else grad_of(b))
        return (dx if ctx.needs_input_grad[0] else None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


class SpaceToDepthFn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.space_to_depth(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.depth_to_space(g.contiguous())


class PixelShuffle2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        return ops.pixel_shuffle2(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.pixel_unshuffle2(g.contiguous())


class BilinearUpFn(Function):
    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return ops.bilinear_up(x.contiguous(), s)

    @staticmethod
    def backward(ctx, g):
        return ops.bilinear_up_bwd(g.contiguous(), ctx.s), None


class PReLUFn(Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.slope = slope
        return ops.prelu(x, slope.detach())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.prelu_bwd(x, ctx.slope.detach(), g.contiguous(), grad_of(ctx.slope)), None


class MaskTokenFn(Function):
    """fea * (1 - w) + mask_token * w  (UNet_arch.py:463-466)."""

    @staticmethod
    def forward(ctx, fea, mask, token):
        mask = mask.to(fea.dtype).contiguous()
        ctx.save_for_backward(mask)
        ctx.token = token
        return ops.mask_token(fea.contiguous(), mask, token.detach().reshape(-1).contiguous())

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return ops.mask_token_bwd(g.contiguous(), mask, grad_of(ctx.token).view(-1)), None, None


# ------------------------------------------------------------------------------------------------------------------
# Bayesian leaves in training (basicsr/bayesian/conv.py:84-114, linear.py:61-90, tools.py:76-84)
# ------------------------------------------------------------------------------------------------------------------
class BayesStep:
    """One training forward's Bayesian bookkeeping: the leaves that drew a weight sample, to be folded back into mu / rho once the
    whole backward pass has run (the sampled weight's gradient is complete only then)."""

    def __init__(self):
        self.leaves = []
        self.done = False
        self.bank = None                 # modules.BayesBank when the samples of this forward were drawn by its one launch

    def finish(self):
        if self.done:
            return
        self.done = True
        if self.bank is not None:        # dmu += dw, drho += dw eps sigmoid(rho) for every tensor of the net in one launch
            ops.bnn_bank_reparam_bwd_(self.bank)
            for m in self.leaves:
                m._ws = m._bs = m._eps_w = m._eps_b = m._sample_owner = None
            return
        for m in self.leaves:
            m.fold_sample_grads()


class BayesAnchorFn(Function):
    """Identity on the network output; its backward (the first node of the pass) queues BayesStep.finish to run after the pass."""

    @staticmethod
    def forward(ctx, out, step):
        ctx.step = step
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        torch.autograd.Variable._execution_engine.queue_callback(ctx.step.finish)
        return g, None


class KLFn(Function):
    """sum over the Bayesian leaves of kl_div(q || EMA prior).mean() for weight (+ bias)  (get_kl_loss, tools.py:76-84)."""

    @staticmethod
    def forward(ctx, leaves, *params):
        out = torch.zeros(1, device=params[0].device, dtype=torch.float32)
        st = getattr(leaves[0], "_sample_owner", None)
        bank = getattr(st, "bank", None)
        if bank is not None and len(bank.leaves) == len(leaves) and all(a is b and a._sample_owner is st for a, b in zip(bank.leaves, leaves)):
            ops.bnn_bank_kl_(bank, out)              # the forward that has just run drew its samples through the bank: same tensors, one launch
            ctx.bank = bank
            return out.reshape(())
        ctx.bank = None
        for m in leaves:
            for mu, rho, pmu, prho in m.kl_terms():
                ops.bnn_kl_(mu.detach(), rho.detach(), pmu, prho, out)
        ctx.leaves = leaves
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        g = g.reshape(1).contiguous().float()
        if ctx.bank is not None:
            ops.bnn_bank_kl_bwd_(ctx.bank, g)
            return (None,) * len(ctx.needs_input_grad)
        for m in ctx.leaves:
            for mu, rho, pmu, prho in m.kl_terms():
                ops.bnn_kl_bwd_(mu.detach(), rho.detach(), pmu, prho, g, grad_of(mu), grad_of(rho))
        return (None,) * (1 + len(ctx.needs_input_grad) - 1)


class ScaledSumFn(Function):
    """a + c * b for two device scalars (l_total = l_pix + 0.01 / mini_batch * l_kl, condition_generator_model.py:185-190)."""

    @staticmethod
    def forward(ctx, a, b, c):
        ctx.c = c
        return ops.add(a.reshape(1).contiguous(), b.reshape(1).contiguous(), c).reshape(())

    @staticmethod
    def backward(ctx, g):
        g1 = g.reshape(1).contiguous()
        return g, ops.add(torch.zeros_like(g1), g1, ctx.c).reshape(()), None


# ------------------------------------------------------------------------------------------------------------------
# DecompDualBranch's bottleneck blocks in training (basicsr/archs/DecompModel_arch.py:57-99)
# ------------------------------------------------------------------------------------------------------------------
class GateAddFn(Function):
    """x_tgt + gate * t  (CrossFusionBlock after its 1x1 transform, :62-66): dt = gate dy, dx_tgt = dy, dgate[c] += sum dy t."""

    @staticmethod
    def forward(ctx, t, gate, x_tgt):
        t, x_tgt = t.contiguous(), x_tgt.contiguous()
        ctx.save_for_backward(t)
        ctx.gate = gate
        return ops.chan_scale(t, gate.detach().reshape(-1).contiguous(), add=x_tgt)

    @staticmethod
    def backward(ctx, dy):
        (t,) = ctx.saved_tensors
        dy = dy.contiguous()
        g = ctx.gate
        ops.chan_dot(dy, t, grad_of(g).view(-1))
        return ops.chan_scale(dy, g.detach().reshape(-1).contiguous()), None, dy


class SEBlockFn(Function):
    """x * sigmoid(W2 relu(W1 mean_hw(x)))  (SEBlock, :68-83)."""

    @staticmethod
    def forward(ctx, x, w1, w2):
        x = x.contiguous()
        y, mean = ops.se_gate(x, w1.detach(), w2.detach(), want_mean=True)
        ctx.save_for_backward(x, y, mean)
        ctx.w = (w1, w2)
        return ops.chan_scale(x, y)

    @staticmethod
    def backward(ctx, dout):
        x, y, mean = ctx.saved_tensors
        w1, w2 = ctx.w
        dout = dout.contiguous()
        dy = ops.chan_dot(dout, x)                                                     # (B,C): gradient of the gate
        dmean = ops.se_gate_bwd(mean, w1.detach(), w2.detach(), y, dy, grad_of(w1), grad_of(w2))
        return ops.chan_scale(dout, y, add_bc=dmean, add_bc_scale=1.0 / x[0, 0].numel()), None, None


class SpatialAttnFn(Function):
    """x * sigmoid(conv_kxk([mean_c x, max_c x]))  (SpatialAttention, :85-99)."""

    @staticmethod
    def forward(ctx, x, w):
        x = x.contiguous()
        out, amap = ops.spatial_attention(x, w.detach(), want_map=True)
        ctx.save_for_backward(x, amap)
        ctx.w = w
        return out

    @staticmethod
    def backward(ctx, dout):
        x, amap = ctx.saved_tensors
        w = ctx.w
        return ops.spatial_attention_bwd(x, dout.contiguous(), amap, w.detach(), grad_of(w)), None


class HamiltonFn(Function):
    """hamilton_product(p, q)[:, 1:] of two (B,4,H,W) maps at full resolution (DecompModel_arch.py:351-352)."""

    @staticmethod
    def forward(ctx, p, q):
        B, _, H, W = p.shape
        q8 = torch.empty(B, 8, H, W, device=p.device, dtype=p.dtype)
        ops.copy_channels(p.contiguous(), q8, 0)
        ops.copy_channels(q.contiguous(), q8, 4)
        ctx.save_for_backward(q8)
        return ops.hamilton(q8)

    @staticmethod
    def backward(ctx, dout):
        (q8,) = ctx.saved_tensors
        d = ops.hamilton_bwd(q8, dout.contiguous())
        B, _, H, W = q8.shape
        dp, dq = torch.empty(B, 4, H, W, device=d.device, dtype=d.dtype), torch.empty(B, 4, H, W, device=d.device, dtype=d.dtype)
        ops.copy_channels(d, dp, 0, src_c0=0, C=4)
        ops.copy_channels(d, dq, 0, src_c0=4, C=4)
        return dp, dq


class Hamilton8Fn(Function):
    """The same on one (B,8,H,W) tensor [p | q] (DecompSingleBranch_arch.py:229-231: the single U-Net's 8 output channels)."""

    @staticmethod
    def forward(ctx, q8):
        q8 = q8.contiguous()
        ctx.save_for_backward(q8)
        return ops.hamilton(q8)

    @staticmethod
    def backward(ctx, dout):
        (q8,) = ctx.saved_tensors
        return ops.hamilton_bwd(q8, dout.contiguous())
