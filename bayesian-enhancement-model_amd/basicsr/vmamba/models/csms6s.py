"""Operator seam of basicsr/vmamba/models/csms6s.py:75-130: the same autograd Function over the same native module
(``selective_scan_cuda_oflex``, here the ctypes binding of libbem_hip.so) called with the reference's argument lists."""
import torch

import selective_scan_cuda_oflex

WITH_SELECTIVESCAN_OFLEX = True


class SelectiveScanCuda(torch.autograd.Function):
    """csms6s.py:75-113 with backend 'oflex' (the only extension the reference builds, kernels/selective_scan/setup.py:40)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, oflex=True, backend=None):
        if backend not in (None, "oflex"):
            raise NotImplementedError(f"selective_scan backend {backend}: only the oflex extension exists on this path")
        ctx.delta_softplus = delta_softplus
        out, x, *rest = selective_scan_cuda_oflex.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, 1, oflex)
        ctx.has = (D is not None, delta_bias is not None)
        ctx.save_for_backward(u, delta, A, B, C, *([D] if D is not None else []), *([delta_bias] if delta_bias is not None else []), x)
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        saved = list(ctx.saved_tensors)
        u, delta, A, B, C = saved[:5]
        x = saved[-1]
        rest = saved[5:-1]
        D = rest.pop(0) if ctx.has[0] else None
        delta_bias = rest.pop(0) if ctx.has[1] else None
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_cuda_oflex.bwd(u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, 1)
        return du, ddelta, dA, dB.to(B.dtype), dC.to(C.dtype), dD, ddelta_bias, None, None, None


SelectiveScanHip = SelectiveScanCuda      # round-1 name


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True, backend=None):
    """Same arguments / result as the reference: u, delta (B,KC,L); A (KC,N); B, C (B,K,N,L); returns y (B,KC,L) in float32
    (oflex) or u.dtype; differentiable.  There is no torch / CPU fallback: ``backend='torch'`` raises."""
    if backend == "torch":
        raise NotImplementedError("selective_scan_fn: the torch fallback of the reference is the CPU oracle's job (oracle/bem_oracle.py), not the product's")
    return SelectiveScanCuda.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, oflex, backend)
