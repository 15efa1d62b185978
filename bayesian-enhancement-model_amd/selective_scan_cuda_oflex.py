"""``selective_scan_cuda_oflex`` -- the native seam of the reference (the module its CUDA extension builds,
kernels/selective_scan/setup.py:98-109; binding selective_scan_oflex.cpp:157-165,245-254,360-363), here a ctypes binding of
libbem_hip.so's C ABI (include/bem_hip.h).  ``basicsr/vmamba/models/csms6s.py`` imports it and calls it exactly like the
reference does (csms6s.py:85,101):

    out, x, *rest = selective_scan_cuda_oflex.fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, 1, oflex)
    du, ddelta, dA, dB, dC, dD, ddelta_bias, *rest = selective_scan_cuda_oflex.bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, 1)

Conventions carried over from the extension: inputs contiguous in the last dimension; A, D, delta_bias float32; u, delta, B, C
of one dtype (float32 / float16 / bfloat16; 16-bit inputs are read as they are by the forward kernel and converted by the library's
cast kernels around the backward one -- arithmetic is float32 throughout, no torch compute op runs at this seam); dim % n_groups == 0; dstate <= 256; outputs freshly
allocated and owned by the caller; errors as RuntimeError; runs on the tensors' device and torch's CURRENT stream, no internal
synchronisation.  ``x`` is the opaque save-for-backward blob (B, D, ceil(L / 2048), 2 N): this implementation recomputes the
chunk states in its backward kernel, so the blob carries nothing and is ignored by ``bwd``.  nrows must be 1 (the only
value the reference passes)."""
import ctypes
import os

import torch

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("BEM_HIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libbem_hip.so"))
        L = ctypes.CDLL(path)                                   # OSError here = the HIP library is missing: there is no fallback
        P, I = ctypes.c_void_p, ctypes.c_int
        L.bem_selective_scan_fwd_f32.argtypes = [P] * 8 + [I] * 6 + [P]
        L.bem_selective_scan_fwd_in16.argtypes = [P] * 8 + [I] * 7 + [P]
        L.bem_cast16_to_f32.argtypes = [P, P, ctypes.c_int64, I, P]
        L.bem_cast_f32_to16.argtypes = [P, P, ctypes.c_int64, I, P]
        L.bem_selective_scan_bwd_f32.argtypes = [P] * 16 + [I] * 6 + [P]
        L.bem_selective_scan_bwd_ws_elems.argtypes = [I] * 4
        L.bem_selective_scan_bwd_ws_elems.restype = ctypes.c_int64
        L.bem_last_error.restype = ctypes.c_char_p
        _LIB = L
    return _LIB


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _check_inputs(u, delta, A, B, C, D, delta_bias):
    for n, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        if not t.is_cuda:
            raise RuntimeError(f"selective_scan_cuda_oflex: {n} must be a CUDA/HIP tensor")       # TORCH_CHECK(u.is_cuda()) :175
    if u.dtype not in (torch.float32, torch.float16, torch.bfloat16) or delta.dtype != u.dtype or B.dtype != u.dtype or C.dtype != u.dtype:
        raise RuntimeError("selective_scan_cuda_oflex: u, delta, B, C must share one of float32 / float16 / bfloat16")
    if A.dtype != torch.float32 or (D is not None and D.dtype != torch.float32) or (delta_bias is not None and delta_bias.dtype != torch.float32):
        raise RuntimeError("selective_scan_cuda_oflex: A, D, delta_bias must be float32")
    if u.stride(-1) != 1 or delta.stride(-1) != 1 or B.stride(-1) != 1 or C.stride(-1) != 1:
        raise RuntimeError("selective_scan_cuda_oflex: inputs must be contiguous in the last dimension")       # :181-182,198-200
    batch, dim, L = u.shape
    n_groups, dstate = B.shape[1], A.shape[1]
    if dim % n_groups or dstate > 256 or tuple(A.shape) != (dim, dstate) or tuple(B.shape) != (batch, n_groups, dstate, L) or C.shape != B.shape \
            or delta.shape != u.shape:
        raise RuntimeError("selective_scan_cuda_oflex: inconsistent shapes")
    return batch, dim, L, n_groups, dstate


_DT16 = {torch.float16: 1, torch.bfloat16: 2}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rc(rc):
    if rc:
        raise RuntimeError(_lib().bem_last_error().decode())


def _f32(t):
    """float32 view of a tensor for the kernels: float32 tensors pass through (made dense if a caller handed a strided one), 16-bit ones go
    through bem_cast16_to_f32."""
    if t is None:
        return None
    t = t.contiguous()
    if t.dtype == torch.float32:
        return t
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    _rc(_lib().bem_cast16_to_f32(_p(t), _p(out), t.numel(), _DT16[t.dtype], _stream()))
    return out


def _to16(t, dtype):
    if dtype == torch.float32:
        return t
    out = torch.empty(t.shape, device=t.device, dtype=dtype)
    _rc(_lib().bem_cast_f32_to16(_p(t), _p(out), t.numel(), _DT16[dtype], _stream()))
    return out


def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows, out_float):
    if nrows != 1:
        raise RuntimeError("selective_scan_cuda_oflex: nrows must be 1")
    batch, dim, L, n_groups, dstate = _check_inputs(u, delta, A, B, C, D, delta_bias)
    Af, Df, bf = (None if t is None else t.contiguous() for t in (A, D, delta_bias))
    out = torch.empty(batch, dim, L, device=u.device, dtype=torch.float32)
    if u.dtype == torch.float32:
        uf, df, Bf, Cf = (t.contiguous() for t in (u, delta, B, C))
        _rc(_lib().bem_selective_scan_fwd_f32(_p(uf), _p(df), _p(Af), _p(Bf), _p(Cf), _p(Df), _p(bf), _p(out), batch, dim, L, dstate, n_groups,
                                              int(bool(delta_softplus)), _stream()))
    else:
        uh, dh, Bh, Ch = (t.contiguous() for t in (u, delta, B, C))
        _rc(_lib().bem_selective_scan_fwd_in16(_p(uh), _p(dh), _p(Af), _p(Bh), _p(Ch), _p(Df), _p(bf), _p(out), _DT16[u.dtype], batch, dim, L, dstate,
                                               n_groups, int(bool(delta_softplus)), _stream()))
    x = torch.empty(batch, dim, (L + 2047) // 2048, 2 * dstate, device=u.device, dtype=torch.float32)
    return [out if out_float else _to16(out, u.dtype), x]


def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
    if nrows != 1:
        raise RuntimeError("selective_scan_cuda_oflex: nrows must be 1")
    batch, dim, L, n_groups, dstate = _check_inputs(u, delta, A, B, C, D, delta_bias)
    if tuple(dout.shape) != (batch, dim, L):
        raise RuntimeError("selective_scan_cuda_oflex: dout shape")
    if dout.dtype not in (torch.float32, torch.float16, torch.bfloat16):
        raise RuntimeError("selective_scan_cuda_oflex: dout must be float32 / float16 / bfloat16")
    uf, df, Af, Bf, Cf, Df, bf, gf = map(_f32, (u, delta, A, B, C, D, delta_bias, dout))
    dev = u.device
    du, dd = torch.empty_like(uf), torch.empty_like(uf)
    dA, dB, dC = torch.empty_like(Af), torch.empty_like(Bf), torch.empty_like(Cf)     # float32 accumulators (:324-332)
    dD = torch.empty_like(Df) if Df is not None else None
    db = torch.empty_like(bf) if bf is not None else None
    ws = torch.empty(int(_lib().bem_selective_scan_bwd_ws_elems(batch, dim, L, dstate)), device=dev, dtype=torch.float32)
    _rc(_lib().bem_selective_scan_bwd_f32(_p(uf), _p(df), _p(Af), _p(Bf), _p(Cf), _p(Df), _p(bf), _p(gf), _p(ws), _p(du), _p(dd), _p(dA), _p(dB), _p(dC),
                                          _p(dD), _p(db), batch, dim, L, dstate, n_groups, int(bool(delta_softplus)), _stream()))
    return [_to16(du, u.dtype), _to16(dd, u.dtype), dA, dB, dC, dD, db]
