"""Operator seam of basicsr/vmamba/models/csms6s.py:75-130 on the HIP selective scan (forward and backward)."""
import torch

from bem import ops


class SelectiveScanHip(torch.autograd.Function):
    """Mirror of SelectiveScanCuda (csms6s.py:75-113): forward saves the inputs, backward returns
    (du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None, None)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, oflex=True, backend=None):
        f = lambda t: None if t is None else t.detach().float().contiguous()
        u_, d_, A_, B_, C_, D_, b_ = map(f, (u, delta, A, B, C, D, delta_bias))
        ctx.delta_softplus = delta_softplus
        ctx.in_dtype = u.dtype
        ctx.save_for_backward(u_, d_, A_, B_, C_, *( [D_] if D_ is not None else []), *([b_] if b_ is not None else []))
        ctx.has = (D_ is not None, b_ is not None)
        y = ops.selective_scan_fwd(u_, d_, A_, B_, C_, D_, b_, delta_softplus)
        return y if oflex else y.to(u.dtype)

    @staticmethod
    def backward(ctx, dout, *args):
        saved = list(ctx.saved_tensors)
        u, delta, A, B, C = saved[:5]
        rest = saved[5:]
        D = rest.pop(0) if ctx.has[0] else None
        bias = rest.pop(0) if ctx.has[1] else None
        du, dd, dA, dB, dC, dD, db = ops.selective_scan_bwd(u, delta, A, B, C, D, bias, dout.float().contiguous(), ctx.delta_softplus)
        return du.to(ctx.in_dtype), dd.to(ctx.in_dtype), dA, dB.to(ctx.in_dtype), dC.to(ctx.in_dtype), dD, db, None, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True, backend=None):
    """Same arguments / result as the reference: u, delta (B,KC,L); A (KC,N); B, C (B,K,N,L); returns y (B,KC,L) in
    float32 (oflex) or u.dtype; differentiable.  ``backend`` is accepted for compatibility; the HIP kernels are the
    only backend (no torch/CPU fallback)."""
    return SelectiveScanHip.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, oflex, backend)
