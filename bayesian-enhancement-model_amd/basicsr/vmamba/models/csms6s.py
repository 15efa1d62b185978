"""Operator seam of basicsr/vmamba/models/csms6s.py:116-130 on the HIP selective scan (forward only)."""
import torch

from bem import ops


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True, backend=None):
    """Same arguments / result as the reference: u, delta (B,KC,L); A (KC,N); B, C (B,K,N,L); returns
    y (B,KC,L) in float32 (oflex) or u.dtype.  ``backend`` is accepted for compatibility; the HIP kernel
    is the only backend (no torch/CPU fallback)."""
    if any(t.requires_grad for t in (u, delta, A, B, C) if t is not None) and torch.is_grad_enabled():
        raise NotImplementedError("selective_scan backward (SURVEY.md row A10) is not built in this round")
    f = lambda t: None if t is None else t.float().contiguous()
    y = ops.selective_scan_fwd(f(u), f(delta), f(A), f(B), f(C), f(D), f(delta_bias), delta_softplus)
    return y if oflex else y.to(u.dtype)
