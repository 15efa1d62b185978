"""Tensor-level wrappers over the C ABI (include/bem_hip.h).

Each wrapper checks operand shapes on the host before anything is launched (a kernel fault can take
the whole GPU host down), allocates the output with torch (device memory is torch's job here) and
launches on torch's current stream.  No wrapper computes anything itself."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import native
from .native import PwArgs, check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32, optional=False):
    if t is None:
        if optional:
            return
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise native.BemNativeError(f"{name} must be a CUDA/HIP tensor: the BEM hot path has no CPU implementation")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


# --------------------------------------------------------------------------- operator seam ----
def selective_scan_fwd(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True):
    """u, delta (Bt,KC,L); A (KC,N); B, C (Bt,K,N,L); D, delta_bias (KC) -> y (Bt,KC,L) f32."""
    for n, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _chk(t, n)
    _chk(D, "D", optional=True)
    _chk(delta_bias, "delta_bias", optional=True)
    Bt, KC, L = u.shape
    if delta.shape != u.shape:
        raise ValueError(f"delta {tuple(delta.shape)} != u {tuple(u.shape)}")
    if B.dim() != 4 or B.shape[0] != Bt or B.shape[3] != L or C.shape != B.shape:
        raise ValueError(f"B/C must be (batch, groups, dstate, L); got {tuple(B.shape)} {tuple(C.shape)}")
    K, N = B.shape[1], B.shape[2]
    if A.shape != (KC, N):
        raise ValueError(f"A must be ({KC},{N}), got {tuple(A.shape)}")
    if KC % K:
        raise ValueError("dim must be a multiple of the number of groups")
    for n, t in (("D", D), ("delta_bias", delta_bias)):
        if t is not None and t.shape != (KC,):
            raise ValueError(f"{n} must be ({KC},)")
    out = torch.empty_like(u)
    check(lib().bem_selective_scan_fwd_f32(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(out),
                                           Bt, KC, L, N, K, int(bool(delta_softplus)), _stream()), "selective_scan_fwd")
    return out


def selective_scan_bwd(u, delta, A, B, C, D, delta_bias, dout, delta_softplus=True):
    """Gradients (du, ddelta, dA, dB, dC, dD|None, ddelta_bias|None) of selective_scan_fwd w.r.t. its inputs."""
    for n, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C), ("dout", dout)):
        _chk(t, n)
    _chk(D, "D", optional=True); _chk(delta_bias, "delta_bias", optional=True)
    Bt, KC, L = u.shape
    K, N = B.shape[1], B.shape[2]
    if delta.shape != u.shape or dout.shape != u.shape or A.shape != (KC, N) or C.shape != B.shape or B.shape[0] != Bt or B.shape[3] != L or KC % K:
        raise ValueError("selective_scan_bwd: shapes")
    du, dd = torch.empty_like(u), torch.empty_like(u)
    dA, dB, dC = torch.empty_like(A), torch.empty_like(B), torch.empty_like(C)
    dD = torch.empty_like(D) if D is not None else None
    db = torch.empty_like(delta_bias) if delta_bias is not None else None
    ws = torch.empty(int(lib().bem_selective_scan_bwd_ws_elems(Bt, KC, L, N)), device=u.device, dtype=torch.float32)
    check(lib().bem_selective_scan_bwd_f32(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(dout), _p(ws), _p(du), _p(dd),
                                           _p(dA), _p(dB), _p(dC), _p(dD), _p(db), Bt, KC, L, N, K, int(bool(delta_softplus)), _stream()),
          "selective_scan_bwd")
    return du, dd, dA, dB, dC, dD, db


def cross_scan(x):
    _chk(x, "x")
    B, C, H, W = x.shape
    xs = torch.empty(B, 4, C, H * W, device=x.device, dtype=x.dtype)
    check(lib().bem_cross_scan_f32(_p(x), _p(xs), B, C, H, W, _stream()), "cross_scan")
    return xs


def cross_merge(ys):
    _chk(ys, "ys")
    B, K, C, H, W = ys.shape
    if K != 4:
        raise ValueError("cross_merge expects 4 directions")
    y = torch.empty(B, C, H * W, device=ys.device, dtype=ys.dtype)
    check(lib().bem_cross_merge_f32(_p(ys), _p(y), B, C, H, W, _stream()), "cross_merge")
    return y


# --------------------------------------------------------------------------- fused SS2D -------
def ss2d_scan(x0, x1, xd0, xd1, dtw, dtb, A, Ds):
    """xd0 / xd1 (B,2,R+2,L): contiguous, or channel slices of a wider (B,*,L) buffer (batch stride taken from the view)."""
    for n, t in (("x0", x0), ("x1", x1), ("dtw", dtw), ("dtb", dtb), ("A", A), ("Ds", Ds)):
        _chk(t, n)
    B, C, L = x0.shape
    R = dtw.shape[2]
    if x1.shape != x0.shape or xd0.shape != (B, 2, R + 2, L) or xd1.shape != xd0.shape:
        raise ValueError(f"ss2d_scan shapes: x {tuple(x0.shape)}/{tuple(x1.shape)} xd {tuple(xd0.shape)}/{tuple(xd1.shape)} R={R}")
    if dtw.shape != (4, C, R) or dtb.shape != (4, C) or A.numel() != 4 * C or Ds.numel() != 4 * C:
        raise ValueError("ss2d_scan parameter shapes")
    bs = []
    for n, t in (("xd0", xd0), ("xd1", xd1)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise native.BemNativeError(f"{n} must be a float32 CUDA/HIP tensor")
        if t.stride()[1:] != ((R + 2) * L, L, 1) or (B > 1 and (t.stride(0) < 2 * (R + 2) * L or (L % 4 == 0 and t.stride(0) % 4))):
            raise ValueError(f"{n}: only the batch stride may differ from a contiguous (B,2,R+2,L) tensor")
        bs.append(t.stride(0) if B > 1 else 0)
    y0 = torch.empty_like(x0)
    y1 = torch.empty_like(x0)
    check(lib().bem_ss2d_scan_strided_f32(_p(x0), _p(x1), _p(xd0), _p(xd1), _p(dtw), _p(dtb), _p(A), _p(Ds), _p(y0), _p(y1),
                                          B, C, L, R, bs[0], bs[1], _stream()), "ss2d_scan")
    return y0, y1


def ss2d_scan_rm_supported(H, W, R):
    return bool(lib().bem_ss2d_scan_rm_supported(H, W, R))


def ss2d_scan_rm(x, xd0, xd1, dtw, dtb, A, Ds):
    """Row-major form: x (B,C,H,W); xd0 (B,2,R+2,L) row-major order, xd1 (B,2,R+2,L) transposed pixel order (either may be
    a batch-strided channel slice); returns y0, y1 both (B,C,H,W) row-major."""
    for n, t in (("x", x), ("dtw", dtw), ("dtb", dtb), ("A", A), ("Ds", Ds)):
        _chk(t, n)
    B, C, H, W = x.shape
    L, R = H * W, dtw.shape[2]
    if not ss2d_scan_rm_supported(H, W, R):
        raise ValueError(f"ss2d_scan_rm: plane {H}x{W} / dt_rank {R} not supported")
    if xd0.shape != (B, 2, R + 2, L) or xd1.shape != xd0.shape or dtw.shape != (4, C, R) or dtb.shape != (4, C) or A.numel() != 4 * C or Ds.numel() != 4 * C:
        raise ValueError("ss2d_scan_rm: shapes")
    bs = []
    for n, t in (("xd0", xd0), ("xd1", xd1)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise native.BemNativeError(f"{n} must be a float32 CUDA/HIP tensor")
        if t.stride()[1:] != ((R + 2) * L, L, 1) or (B > 1 and (t.stride(0) < 2 * (R + 2) * L or (L % 4 == 0 and t.stride(0) % 4))):
            raise ValueError(f"{n}: only the batch stride may differ from a contiguous (B,2,R+2,L) tensor")
        bs.append(t.stride(0) if B > 1 else 0)
    y0, y1 = torch.empty_like(x), torch.empty_like(x)
    check(lib().bem_ss2d_scan_rm_f32(_p(x), _p(xd0), _p(xd1), _p(dtw), _p(dtb), _p(A), _p(Ds), _p(y0), _p(y1), B, C, H, W, R, bs[0], bs[1], _stream()),
          "ss2d_scan_rm")
    return y0, y1


def transpose_plane_slice(x, c0, C):
    """x (B,Ct,H,W) contiguous -> transposed planes of channels [c0, c0+C): (B,C,W,H) contiguous."""
    _chk(x, "x")
    B, Ct, H, W = x.shape
    if c0 < 0 or c0 + C > Ct or B * C > 65535:
        raise ValueError("transpose_plane_slice: channel range / plane count")
    out = torch.empty(B, C, W, H, device=x.device, dtype=x.dtype)
    check(lib().bem_transpose_planes_f32(ctypes.c_void_p(x.data_ptr() + 4 * c0 * H * W), Ct * H * W, _p(out), C * H * W, B, C, H, W, _stream()),   # (nbatch, planes per batch)
          "transpose_planes")
    return out


# --------------------------------------------------------------------------- pointwise GEMM ---
# Two packed operand formats: "x6" (three bf16 limbs per f32 value, six limb products on v_mfma_f32_32x32x16_bf16, bem_pw_gemm_x6_f32 --
# f32-level error at 6/16 of the f32 matrix-pipe cost), what every pointwise GEMM consumes; and "f32" (v_mfma_f32_32x32x2_f32 operand
# order, pack_pw_weight(x6=False)), the weight format of the implicit-GEMM convolutions (bem_conv2d_mfma_f32).
USE_X6 = True          # the pointwise GEMMs run on the bf16-limb kernels only (the f32-MFMA GEMMs were removed in round 3)
# Bumped whenever parameters are rewritten behind torch's back (the fused optimizer step, bem.train.BemAdamW): every cache of
# derived weights (packed / transposed / flipped copies) is keyed on it as well as on the tensors' data_ptr / _version.
WEIGHT_EPOCH = [0]


def bump_weight_epoch():
    WEIGHT_EPOCH[0] += 1


def tensor_version(t) -> int:
    """t._version for cache keys; tensors created under torch.inference_mode() (the reference's eval.py runs there) carry no
    version counter and cannot be modified in place later, so a constant serves."""
    return -1 if t.is_inference() else t._version


def packed_elems(M: int, K: int, x6: bool = False) -> int:
    return int(lib().bem_pw_x6_packed_elems(M, K) if x6 else lib().bem_pw_packed_elems(M, K))


def pack_pw_weight(W, x6=None):
    """(M,K) or (nsets,M,K) natural weights -> packed MFMA operand order, shape (nsets, packed).
    x6: None = module default (USE_X6), False = f32 operands (also what the MFMA convolution consumes)."""
    if W.dim() == 2:
        W = W[None]
    ns, M, K = W.shape
    x6 = USE_X6 if x6 is None else x6
    out = torch.empty(ns, packed_elems(M, K, x6), device=W.device, dtype=W.dtype)
    if x6 and not W.is_contiguous():            # a transposed / sliced view (W^T of an input-gradient GEMM): packed where it lies
        if not W.is_cuda or W.dtype != torch.float32:
            raise native.BemNativeError("pack_pw_weight: a float32 CUDA/HIP tensor")
        check(lib().bem_pack_pw_weight_x6_strided(_p(W), _p(out), ns, M, K, W.stride(0), W.stride(1), W.stride(2), _stream()), "pack_pw_weight")
        out._bem_mk = (M, K)
        return out
    _chk(W, "W")
    fn = lib().bem_pack_pw_weight_x6 if x6 else lib().bem_pack_pw_weight_f32
    check(fn(_p(W), _p(out), ns, M, K, _stream()), "pack_pw_weight")
    out._bem_mk = (M, K)          # the x6 size only pins ceil(K/16): keep the exact logical shape for pw_gemm's check
    return out


def pw_gemm(x1, Wp, M, *, x2=None, in_mode=0, ln=None, ln_eps=1e-5, bias=None, res=None, prelu=None,
            convT_Win: int = 0, out=None):
    """x1 (B,C1,*spatial); Wp packed (1|B, packed(M,K)); returns (B,M,*spatial) (or (B,M/4,2H,2W) for convT)."""
    _chk(x1, "x1"); _chk(Wp, "Wp"); _chk(x2, "x2", optional=True); _chk(bias, "bias", optional=True)
    _chk(res, "res", optional=True); _chk(prelu, "prelu", optional=True)
    B, C1 = x1.shape[0], x1.shape[1]
    sp = tuple(x1.shape[2:])
    L = 1
    for s in sp:
        L *= s
    C2 = 0
    if in_mode:
        if x2 is None or x2.shape[0] != B or tuple(x2.shape[2:]) != sp:
            raise ValueError("pw_gemm: x2 shape")
        C2 = x2.shape[1]
    K = C1 + C2 if in_mode == 2 else C1
    if in_mode == 1 and C1 != C2:
        raise ValueError("pw_gemm sum mode: channel mismatch")
    if getattr(Wp, "_bem_mk", (M, K)) != (M, K):
        raise ValueError(f"pw_gemm: weights packed for (M, K) = {Wp._bem_mk}, called with M={M} K={K}")
    if Wp.dim() != 2 or Wp.shape[1] != packed_elems(M, K, True) or Wp.shape[0] not in (1, B):
        raise ValueError(f"pw_gemm: packed weight shape {tuple(Wp.shape)} does not match M={M} K={K} B={B} (x6 format, pack_pw_weight)")
    a = PwArgs()
    a.x1, a.x2, a.C1, a.C2, a.in_mode = x1.data_ptr(), (x2.data_ptr() if x2 is not None else 0), C1, C2, in_mode
    if ln is not None:
        lw, lb = ln
        _chk(lw, "ln_w"); _chk(lb, "ln_b")
        if lw.numel() != K or lb.numel() != K:
            raise ValueError("pw_gemm: LayerNorm size != K")
        a.ln_w, a.ln_b = lw.data_ptr(), lb.data_ptr()
    a.ln_eps = ln_eps
    a.Wp, a.w_bstride = Wp.data_ptr(), (Wp.shape[1] if Wp.shape[0] > 1 else 0)
    if bias is not None:
        if bias.shape[-1] != M or (bias.dim() == 2 and bias.shape[0] not in (1, B)):
            raise ValueError("pw_gemm: bias shape")
        a.bias, a.bias_bstride = bias.data_ptr(), (M if (bias.dim() == 2 and bias.shape[0] > 1) else 0)
    if prelu is not None:
        if prelu.numel() != 1:
            raise ValueError("pw_gemm: PReLU with one slope only")
        a.prelu, a.act = prelu.data_ptr(), 1
    if convT_Win:
        if len(sp) != 2 or sp[1] != convT_Win or M % 4 or res is not None:
            raise ValueError("pw_gemm convT mode: bad arguments")
        oshape = (B, M // 4, 2 * sp[0], 2 * sp[1])
        a.out_mode, a.Win = 1, convT_Win
    else:
        oshape = (B, M) + sp
    if res is not None:
        if tuple(res.shape) != oshape:
            raise ValueError(f"pw_gemm: residual shape {tuple(res.shape)} != {oshape}")
        a.res = res.data_ptr()
    if out is None:
        out = torch.empty(oshape, device=x1.device, dtype=x1.dtype)
    elif tuple(out.shape) != oshape or not out.is_contiguous():
        raise ValueError("pw_gemm: out shape")
    a.out = out.data_ptr()
    a.B, a.M, a.K, a.L = B, M, K, L
    check(lib().bem_pw_gemm_x6_f32(ctypes.byref(a), _stream()), "pw_gemm")
    return out


def empty_padded(shape, device, pad=4):
    """Contiguous float32 tensor of ``shape`` that may be read ``pad`` elements before its first and after its last element
    (it is a view into a larger allocation; the 16-byte alignment of a fresh allocation is kept for pad % 4 == 0)."""
    n = 1
    for s_ in shape:
        n *= s_
    buf = torch.empty(n + 2 * pad, device=device, dtype=torch.float32)
    return buf[pad:pad + n].view(shape)


# SS2D front half (LayerNorm + in_proj + depthwise 3x3 + SiLU + x_proj) in one kernel for C <= 48, deterministic weights.  BEM_SS2D_FRONT=0
# restores the three-kernel chain.
SS2D_FRONT = os.environ.get("BEM_SS2D_FRONT", "1") != "0"


def ss2d_front_supported(C, Mx):
    return USE_X6 and SS2D_FRONT and C <= 48 and C % 8 == 0 and Mx <= 32


def ss2d_front(x, ln_w, ln_b, ln_eps, Wp_in, bias_in, dww, dwb, Wp_x, Mx):
    """xc = SiLU(dw3x3(in_proj(LayerNorm2d(x)))) (B,C,H,W) and xd = x_proj(xc) (B,Mx,H,W) in ONE kernel (bem_ss2d_front_x6_f32).
    Wp_in = pack_pw_weight(W_in (C,C), x6=True), Wp_x = pack_pw_weight(W_x (Mx,C), x6=True); dww (C,1,3,3) or (C,9)."""
    for n_, t in (("x", x), ("ln_w", ln_w), ("ln_b", ln_b), ("Wp_in", Wp_in), ("dww", dww), ("Wp_x", Wp_x)):
        _chk(t, n_)
    _chk(bias_in, "bias_in", optional=True); _chk(dwb, "dwb", optional=True)
    B, C, H, W = x.shape
    if not (USE_X6 and C <= 48 and C % 8 == 0 and Mx <= 32):
        raise ValueError(f"ss2d_front: C = {C} (<= 48, % 8) / Mx = {Mx} (<= 32) not supported")
    if ln_w.numel() != C or ln_b.numel() != C or dww.numel() != 9 * C or (dwb is not None and dwb.numel() != C) or (bias_in is not None and bias_in.numel() != C):
        raise ValueError("ss2d_front: parameter shapes")
    for Wp, (M, K), nm in ((Wp_in, (C, C), "Wp_in"), (Wp_x, (Mx, C), "Wp_x")):
        if Wp.dim() != 2 or Wp.shape[0] != 1 or Wp.shape[1] != packed_elems(M, K, True) or getattr(Wp, "_bem_mk", (M, K)) != (M, K):
            raise ValueError(f"ss2d_front: {nm} {tuple(Wp.shape)} does not match M={M} K={K} (x6 format, one weight set)")
    xc = torch.empty_like(x)
    xd = torch.empty(B, Mx, H, W, device=x.device, dtype=x.dtype)
    check(lib().bem_ss2d_front_x6_f32(_p(x), _p(ln_w), _p(ln_b), float(ln_eps), _p(Wp_in), _p(bias_in), _p(dww), _p(dwb), _p(Wp_x), _p(xc), _p(xd),
                                      B, C, Mx, H, W, _stream()), "ss2d_front")
    return xc, xd


def gate_interleave(Hd, device):
    """row permutation of a (2Hd, ...) project_in parameter into the gate-interleaved order of bem_gdmlp_x6_f32:
    new row 32 j + 2 c + s = old row s Hd + 16 j + c."""
    j = torch.arange(Hd // 16, device=device)[:, None, None]
    c = torch.arange(16, device=device)[None, :, None]
    s_ = torch.arange(2, device=device)[None, None, :]
    return (s_ * Hd + 16 * j + c).reshape(-1)


# the fully fused gdMlp branch (bem_gdmlp_x6_f32): C <= GDMLP_X6_MAXC, deterministic weights.  BEM_GDMLP_X6=0 restores the chains.
GDMLP_X6 = os.environ.get("BEM_GDMLP_X6", "1") != "0"
GDMLP_X6_MAXC = int(os.environ.get("BEM_GDMLP_X6_MAXC", "80"))


def gdmlp_x6_supported(C, Hd):
    return USE_X6 and GDMLP_X6 and C <= GDMLP_X6_MAXC and Hd % 16 == 0


def dw_gate_params10(dww, dwb, Hd):
    """depthwise (2Hd,1,3,3) / (2Hd)|None parameters -> the (Hd,10,2) block of bem_gdmlp_x6_f32: per gate channel nine (w1, w2) tap pairs
    and the (b1, b2) bias pair (zeros without a bias) -- 80 bytes per channel, moved chunk-wise into LDS by the kernel."""
    w = dww.reshape(2, Hd, 9).permute(1, 2, 0)
    b = torch.zeros(Hd, 1, 2, device=dww.device, dtype=dww.dtype) if dwb is None else dwb.reshape(2, Hd).t().reshape(Hd, 1, 2)
    return torch.cat([w, b], 1).contiguous()


def gdmlp_x6(x, ln_w, ln_b, ln_eps, Wp_gate, bias_gate, dw10, Wp_out, bias_out, Hd):
    """x + project_out(GELU(h1) * h2) + b_o with [h1; h2] = dw3x3(project_in(LayerNorm2d(x))) in ONE kernel (bem_gdmlp_x6_f32).
    Wp_gate = pack_pw_weight(W_i[gate_interleave(Hd)], x6=True), bias_gate = b_i[gate_interleave(Hd)] (zeros without a bias);
    dw10 = dw_gate_params10(...); Wp_out = pack_pw_weight(W_o, x6=True)."""
    for n_, t in (("x", x), ("ln_w", ln_w), ("ln_b", ln_b), ("Wp_gate", Wp_gate), ("bias_gate", bias_gate), ("dw10", dw10), ("Wp_out", Wp_out)):
        _chk(t, n_)
    _chk(bias_out, "bias_out", optional=True)
    B, C, H, W = x.shape
    if not (USE_X6 and C <= 80 and Hd % 16 == 0):
        raise ValueError(f"gdmlp_x6: C = {C} (<= 80) / Hd = {Hd} (% 16) not supported")
    if tuple(dw10.shape) != (Hd, 10, 2):
        raise ValueError("gdmlp_x6: depthwise parameters must come from dw_gate_params10")
    if ln_w.numel() != C or ln_b.numel() != C or bias_gate.numel() != 2 * Hd or (bias_out is not None and bias_out.numel() != C):
        raise ValueError("gdmlp_x6: parameter shapes")
    for Wp, (M, K), nm in ((Wp_gate, (2 * Hd, C), "Wp_gate"), (Wp_out, (C, Hd), "Wp_out")):
        if Wp.dim() != 2 or Wp.shape[0] != 1 or Wp.shape[1] != packed_elems(M, K, True) or getattr(Wp, "_bem_mk", (M, K)) != (M, K):
            raise ValueError(f"gdmlp_x6: {nm} {tuple(Wp.shape)} does not match M={M} K={K} (x6 format, one weight set)")
    out = torch.empty_like(x)
    s = _timed("gdmlp_x6", 8.0 * x.numel(), 6.0 * 2.0 * B * H * W * (2 * Hd * C + Hd * C)) if _PROF is not None and _PROF["kernel"] == "gdmlp_x6" and _PROF["pred"](C) else None
    check(lib().bem_gdmlp_x6_f32(_p(x), _p(ln_w), _p(ln_b), float(ln_eps), _p(Wp_gate), _p(bias_gate), _p(dw10), _p(Wp_out),
                                 _p(bias_out), _p(out), B, C, Hd, H, W, _stream()), "gdmlp_x6")
    _timed_end(s)
    return out


def dwconv3x3(x, w, bias=None, mode=0):
    """mode 0 plain, 1 SiLU, 2 gdMlp gate (x has 2*Cout channels), 3 PostSmooth.  w (Cw,1,3,3) or (B,Cw,1,3,3)."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias", optional=True)
    B, Cin, H, W = x.shape
    Cout = Cin // 2 if mode == 2 else Cin
    per_b = w.dim() == 5
    Cw = w.shape[1] if per_b else w.shape[0]
    if Cw != Cin or tuple(w.shape[-3:]) != (1, 3, 3) or (per_b and w.shape[0] != B):
        raise ValueError(f"dwconv3x3: weight {tuple(w.shape)} vs input channels {Cin}")
    bb = 0
    if bias is not None:
        if bias.shape[-1] != Cin:
            raise ValueError("dwconv3x3: bias shape")
        bb = Cin if (bias.dim() == 2 and bias.shape[0] == B and B > 1) else 0
        if bias.dim() == 2 and bias.shape[0] not in (1, B):
            raise ValueError("dwconv3x3: bias batch")
    out = torch.empty(B, Cout, H, W, device=x.device, dtype=x.dtype)
    check(lib().bem_dwconv3x3_f32(_p(x), _p(w), (Cin * 9 if per_b else 0), _p(bias), bb, _p(out), B, Cout, H, W, mode,
                                  _stream()), "dwconv3x3")
    return out


_CONV_PACK = {}          # (data_ptr, version, shape) -> packed (Cout, Cin*KH*KW) weight for the MFMA conv
USE_CONV_MFMA = __import__("os").environ.get("BEM_CONV_MFMA", "1") != "0"


def _packed_conv_weight(w):
    key = (w.data_ptr(), tensor_version(w), tuple(w.shape), WEIGHT_EPOCH[0])
    hit = _CONV_PACK.get(key)
    if hit is None:
        if len(_CONV_PACK) > 256:
            _CONV_PACK.clear()
        # the source tensor is kept alive with its packed copy: a freed-and-reused address must never alias a stale entry
        hit = (w, pack_pw_weight(w.reshape(w.shape[0], -1).contiguous(), x6=False))
        _CONV_PACK[key] = hit
    return hit[1]


_CONV_PACK_X6 = {}       # same keys -> (9 taps, x6-packed (Cout, Cin)) weights for the shifted-tap 3x3 convolution
USE_CONV_X6 = __import__("os").environ.get("BEM_CONV_X6", "1") != "0"
# the 16-tap form of the 4x4 stride-2 convolution measured no faster than the f32-MFMA im2col kernel (504 vs 537 us at 40 -> 80): off by default
# 4x4 stride-2 down-sampling convs on the x6 matrix-core kernel with coalesced row loads and LDS-staged tap weights (conv4_x6.hip);
# BEM_CONV4_FAST=0 restores the f32-MFMA im2col kernel
CONV4_FAST = __import__("os").environ.get("BEM_CONV4_FAST", "1") != "0"


def _packed_conv_weight_x6(w):
    key = (w.data_ptr(), tensor_version(w), tuple(w.shape), WEIGHT_EPOCH[0])
    hit = _CONV_PACK_X6.get(key)
    if hit is None:
        if len(_CONV_PACK_X6) > 256:
            _CONV_PACK_X6.clear()
        taps = w.permute(2, 3, 0, 1).reshape(w.shape[2] * w.shape[3], w.shape[0], w.shape[1]).contiguous()       # tap = ky * KW + kx
        hit = (w, pack_pw_weight(taps, x6=True))
        _CONV_PACK_X6[key] = hit
    return hit[1]


def conv2d(x, w, bias=None, stride=1, pad=1, relu=False, res1=None, res2=None, cin_slice=None, dilation=1):
    """Dense conv.  ``cin_slice=(c0, Cin)`` convolves channels [c0, c0+Cin) of a wider contiguous x.
    Runs as an implicit GEMM on the matrix cores (Cout <= 160), else on the direct VALU kernel.
    dilation 2 (3x3, pad 2) and 3x3 stride 2 exist in the shifted-tap form only (QD model2 / model3)."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias", optional=True)
    _chk(res1, "res1", optional=True); _chk(res2, "res2", optional=True)
    B, Ct, H, W = x.shape
    Cout, Cin, KH, KW = w.shape
    c0 = 0
    if cin_slice is not None:
        c0, cs = cin_slice
        if cs != Cin or c0 + Cin > Ct:
            raise ValueError("conv2d: channel slice out of range")
    elif Ct != Cin:
        raise ValueError(f"conv2d: input has {Ct} channels, weight expects {Cin}")
    Ho, Wo = (H + 2 * pad - dilation * (KH - 1) - 1) // stride + 1, (W + 2 * pad - dilation * (KW - 1) - 1) // stride + 1
    out = torch.empty(B, Cout, Ho, Wo, device=x.device, dtype=x.dtype)
    if dilation != 1 or (KH, KW, stride) == (3, 3, 2):
        if (KH, KW) != (3, 3) or pad != dilation or not USE_X6 or Wo % 2 or Cin % 8 or (c0 * H * W) % 2:
            raise ValueError("conv2d: dilated / strided 3x3 convolutions run in the tap form only (pad = dilation, even output width, Cin % 8 == 0)")
        for n, r in (("res1", res1), ("res2", res2)):
            if r is not None and r.shape != out.shape:
                raise ValueError(f"conv2d: {n} shape")
        xp_ = ctypes.c_void_p(x.data_ptr() + 4 * c0 * H * W)
        check(lib().bem_conv_taps_x6_f32(xp_, Ct * H * W, _p(_packed_conv_weight_x6(w)), _p(bias), _p(res1), _p(res2), _p(out), B, Cin, H, W, Cout, 3, stride,
                                         dilation, int(relu), _stream()), "conv_taps_x6")
        return out
    for n, r in (("res1", res1), ("res2", res2)):
        if r is not None and r.shape != out.shape:
            raise ValueError(f"conv2d: {n} shape")
    if bias is not None and bias.shape != (Cout,):
        raise ValueError("conv2d: bias shape")
    xp = ctypes.c_void_p(x.data_ptr() + 4 * c0 * H * W)
    if USE_CONV_X6 and USE_X6 and (KH, KW, stride, pad) == (3, 3, 1, 1) and W % 2 == 0 and Cin % 8 == 0 and (c0 * H * W) % 2 == 0:
        # nine shifted 1x1 taps on the bf16-limb GEMM machinery (pw_gemm_x6.hip)
        check(lib().bem_conv3x3_x6_f32(xp, Ct * H * W, _p(_packed_conv_weight_x6(w)), _p(bias), _p(res1), _p(res2), _p(out), B, Cin, H, W,
                                       Cout, int(relu), _stream()), "conv3x3_x6")
        return out
    conv4_fast = CONV4_FAST and USE_X6 and (KH, KW, stride, pad) == (4, 4, 2, 1) and res1 is None and res2 is None and (c0 * H * W) % 4 == 0 \
        and (Ct * H * W) % 4 == 0 and x.data_ptr() % 16 == 0 and lib().bem_conv4x4s2_fast_supported(Cin, H, W) == 1
    if conv4_fast:
        # the coalesced-row form (conv_rows_x6.hip; power-of-two output widths <= 64); other shapes: the f32-MFMA implicit GEMM below
        check(lib().bem_conv4x4s2_x6_f32(xp, Ct * H * W, _p(_packed_conv_weight_x6(w)), _p(bias), _p(res1), _p(res2), _p(out), B, Cin, H, W,
                                         Cout, int(relu), _stream()), "conv4x4s2_x6")
        return out
    if USE_CONV_MFMA and Cout <= 160 and ((KH, KW, stride) in ((3, 3, 1), (4, 4, 2))):
        check(lib().bem_conv2d_mfma_f32(xp, Ct * H * W, _p(_packed_conv_weight(w)), _p(bias), _p(res1), _p(res2), _p(out), B, Cin, H, W,
                                        Cout, KH, KW, stride, pad, int(relu), _stream()), "conv2d_mfma")
        return out
    check(lib().bem_conv2d_f32(xp, Ct * H * W, _p(w), _p(bias), _p(res1), _p(res2), _p(out), B, Cin, H, W, Cout, KH, KW,
                               stride, pad, int(relu), _stream()), "conv2d")
    return out


# --------------------------------------------------------------------------- quaternion / Haar -
def quat_dwt(x, c0=0):
    """x (B,Ct,H,W); channels [c0,c0+3) are RGB -> (B,32,H/2,W/2)."""
    _chk(x, "x")
    B, Ct, H, W = x.shape
    if c0 + 3 > Ct or H % 2 or W % 2:
        raise ValueError("quat_dwt: needs 3 channels and even H, W")
    out = torch.empty(B, 32, H // 2, W // 2, device=x.device, dtype=x.dtype)
    check(lib().bem_quat_dwt_f32(ctypes.c_void_p(x.data_ptr() + 4 * c0 * H * W), Ct * H * W, _p(out), B, H, W, _stream()), "quat_dwt")
    return out


def dwt(x):
    _chk(x, "x")
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise ValueError("dwt: even H, W required")
    out = torch.empty(B, 4 * C, H // 2, W // 2, device=x.device, dtype=x.dtype)
    check(lib().bem_dwt_f32(_p(x), _p(out), B, C, H, W, _stream()), "dwt")
    return out


def iwt(x):
    _chk(x, "x")
    B, C4, H, W = x.shape
    if C4 % 4:
        raise ValueError("iwt: channels must be a multiple of 4")
    out = torch.empty(B, C4 // 4, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
    check(lib().bem_iwt_f32(_p(x), _p(out), B, C4, H, W, _stream()), "iwt")
    return out


def iwt_hamilton(q1w, q2w):
    _chk(q1w, "q1w"); _chk(q2w, "q2w")
    B, C, h, w = q1w.shape
    if C != 16 or q2w.shape != q1w.shape:
        raise ValueError("iwt_hamilton: expects two (B,16,h,w) tensors")
    out = torch.empty(B, 3, 2 * h, 2 * w, device=q1w.device, dtype=q1w.dtype)
    check(lib().bem_iwt_hamilton_f32(_p(q1w), _p(q2w), _p(out), B, h, w, _stream()), "iwt_hamilton")
    return out


def hamilton(q):
    _chk(q, "q")
    B, C, H, W = q.shape
    if C != 8:
        raise ValueError("hamilton: expects (B,8,H,W)")
    out = torch.empty(B, 3, H, W, device=q.device, dtype=q.dtype)
    check(lib().bem_hamilton_f32(_p(q), _p(out), B, H, W, _stream()), "hamilton")
    return out


def hamilton_bwd(q8, dout):
    """Backward of hamilton(): q8 (B,8,H,W), dout (B,3,H,W) -> (B,8,H,W) = [dp | dq]."""
    _chk(q8, "q8"); _chk(dout, "dout")
    B, C, H, W = q8.shape
    if C != 8 or tuple(dout.shape) != (B, 3, H, W):
        raise ValueError("hamilton_bwd: shapes")
    d = torch.empty_like(q8)
    check(lib().bem_hamilton_bwd_f32(_p(q8), _p(dout), _p(d), B, H, W, _stream()), "hamilton_bwd")
    return d


def hamilton_full(q1, q2):
    """QD/quaternion.py hamilton_product: (B,4,H,W) x (B,4,H,W) -> (B,4,H,W) = [real, i, j, k]."""
    _chk(q1, "q1"); _chk(q2, "q2")
    B, C, H, W = q1.shape
    if C != 4 or q2.shape != q1.shape:
        raise ValueError("hamilton_full: expects two (B,4,H,W) tensors")
    out = torch.empty_like(q1)
    check(lib().bem_hamilton_full_f32(_p(q1), _p(q2), _p(out), B, H, W, _stream()), "hamilton_full")
    return out


def attn_fold(f1, f2, attn_w, fuse_w, fuse_b):
    """Channel cross attention + fuse conv folded into per-image (32x64) weights: returns (Wp (B, x6-packed(32,64)), bias (B,32))."""
    for n, t in (("f1", f1), ("f2", f2), ("attn_w", attn_w), ("fuse_w", fuse_w), ("fuse_b", fuse_b)):
        _chk(t, n)
    B, C, H, W = f1.shape
    if C != 32 or f2.shape != f1.shape or attn_w.numel() != 8 * (32 * 32 + 32) or fuse_w.numel() != 32 * 64 or fuse_b.numel() != 32:
        raise ValueError("attn_fold: shapes")
    L = H * W
    stats = torch.empty(B, 32 * 32 + 64, device=f1.device, dtype=torch.float64)
    check(lib().bem_attn_stats_f64(_p(f1), _p(f2), _p(stats), B, L, _stream()), "attn_stats")
    Wn = torch.empty(B, 32, 64, device=f1.device, dtype=torch.float32)
    bias = torch.empty(B, 32, device=f1.device, dtype=torch.float32)
    check(lib().bem_attn_fold_f32(_p(stats), _p(attn_w), _p(fuse_w), _p(fuse_b), _p(Wn), _p(bias), B, L, _stream()), "attn_fold")
    return pack_pw_weight(Wn, x6=True), bias


# --------------------------------------------------------------------------- layout helpers ---
def transpose_planes(x):
    """(B,C,H,W) -> (B,C,W,H), contiguous."""
    _chk(x, "x")
    B, C, H, W = x.shape
    out = torch.empty(B, C, W, H, device=x.device, dtype=x.dtype)
    n = B * C
    # plane count goes into gridDim.z (<= 65535): split large batches
    step = 65535
    xf, of = x.view(n, H * W), out.view(n, H * W)
    for s in range(0, n, step):
        m = min(step, n - s)
        check(lib().bem_transpose_planes_f32(_p(xf[s:]), 0, _p(of[s:]), 0, 1, m, H, W, _stream()), "transpose_planes")
    return out


def copy_channels(src, dst, dst_c0, src_c0=0, C=None):
    """dst[:, dst_c0:dst_c0+C] = src[:, src_c0:src_c0+C] (same spatial size)."""
    _chk(src, "src"); _chk(dst, "dst")
    B, Cs = src.shape[0], src.shape[1]
    C = Cs - src_c0 if C is None else C
    L = src[0, 0].numel()
    if dst.shape[0] != B or dst[0, 0].numel() != L or dst_c0 + C > dst.shape[1] or src_c0 + C > Cs:
        raise ValueError("copy_channels: shapes")
    check(lib().bem_copy_channels_f32(ctypes.c_void_p(src.data_ptr() + 4 * src_c0 * L), Cs * L,
                                      ctypes.c_void_p(dst.data_ptr() + 4 * dst_c0 * L), dst.shape[1] * L, B, C, L,
                                      _stream()), "copy_channels")
    return dst


def copy_channels_rep(src, dst, dst_c0, rep, src_c0=0, C=None):
    """dst[b, dst_c0:dst_c0+C] = src[b // rep, src_c0:src_c0+C]  (dst has rep times the rows of src)."""
    _chk(src, "src"); _chk(dst, "dst")
    Bs, Cs = src.shape[0], src.shape[1]
    C = Cs - src_c0 if C is None else C
    L = src[0, 0].numel()
    if dst.shape[0] != Bs * rep or dst[0, 0].numel() != L or dst_c0 + C > dst.shape[1] or src_c0 + C > Cs:
        raise ValueError("copy_channels_rep: shapes")
    check(lib().bem_copy_channels_rep_f32(ctypes.c_void_p(src.data_ptr() + 4 * src_c0 * L), Cs * L,
                                          ctypes.c_void_p(dst.data_ptr() + 4 * dst_c0 * L), dst.shape[1] * L, Bs * rep, C, L, int(rep),
                                          _stream()), "copy_channels_rep")
    return dst


def add_channels(src, dst, dst_c0, src_c0=0, C=None):
    """dst[:, dst_c0:dst_c0+C] += src[:, src_c0:src_c0+C] (same spatial size)."""
    _chk(src, "src"); _chk(dst, "dst")
    B, Cs = src.shape[0], src.shape[1]
    C = Cs - src_c0 if C is None else C
    L = src[0, 0].numel()
    if dst.shape[0] != B or dst[0, 0].numel() != L or dst_c0 + C > dst.shape[1] or src_c0 + C > Cs:
        raise ValueError("add_channels: shapes")
    check(lib().bem_add_channels_f32(ctypes.c_void_p(src.data_ptr() + 4 * src_c0 * L), Cs * L,
                                     ctypes.c_void_p(dst.data_ptr() + 4 * dst_c0 * L), dst.shape[1] * L, B, C, L,
                                     _stream()), "add_channels")
    return dst


def bilinear_up(src, s, dst=None, dst_c0=0):
    _chk(src, "src")
    B, C, H, W = src.shape
    if dst is None:
        dst = torch.empty(B, C, H * s, W * s, device=src.device, dtype=src.dtype)
        dst_c0 = 0
    _chk(dst, "dst")
    if dst.shape[0] != B or dst.shape[2] != H * s or dst.shape[3] != W * s or dst_c0 + C > dst.shape[1]:
        raise ValueError("bilinear_up: dst shape")
    Lo = H * s * W * s
    check(lib().bem_bilinear_up_f32(_p(src), C * H * W, ctypes.c_void_p(dst.data_ptr() + 4 * dst_c0 * Lo),
                                    dst.shape[1] * Lo, B, C, H, W, s, _stream()), "bilinear_up")
    return dst


def space_to_depth(x):
    _chk(x, "x")
    B, C, H, W = x.shape
    if H % 2 or W % 2:
        raise ValueError("space_to_depth: even H, W required")
    out = torch.empty(B, 4 * C, H // 2, W // 2, device=x.device, dtype=x.dtype)
    check(lib().bem_space_to_depth_f32(_p(x), _p(out), B, C, H, W, _stream()), "space_to_depth")
    return out


def pixel_shuffle2(x):
    _chk(x, "x")
    B, C4, H, W = x.shape
    if C4 % 4:
        raise ValueError("pixel_shuffle2: channels % 4")
    out = torch.empty(B, C4 // 4, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
    check(lib().bem_pixel_shuffle2_f32(_p(x), _p(out), B, C4 // 4, H, W, _stream()), "pixel_shuffle2")
    return out


# ------------------------------------------------------------------- Stage-I training pieces ----
def store_words(dst, host, n):
    """dst[:n] (device, 32-bit elements) = host[:n] (CPU tensor of a 32-bit dtype), n <= 512, as the arguments of one launch."""
    _chk(dst, "dst")
    if host.is_cuda or host.element_size() != 4 or not host.is_contiguous() or n > host.numel() or n > dst.numel():
        raise ValueError("store_words: a contiguous CPU tensor of 32-bit elements")
    check(lib().bem_store_words(_p(dst), ctypes.c_void_p(host.data_ptr()), int(n), _stream()), "store_words")


def bnn_prior_ema_(prior_mu, prior_rho, mu, rho, decay, decay_dev=None):
    for t, nm in ((prior_mu, "prior_mu"), (prior_rho, "prior_rho"), (mu, "mu"), (rho, "rho")):
        _chk(t, nm)
    if not (prior_mu.shape == prior_rho.shape == mu.shape == rho.shape):
        raise ValueError("bnn_prior_ema_: shapes differ")
    check(lib().bem_bnn_prior_ema_f32(_p(prior_mu), _p(prior_rho), _p(mu), _p(rho), float(decay), _p(decay_dev), mu.numel(), _stream()), "bnn_prior_ema")


def bnn_bank_sample(bank, decay, decay_dev, seed, stream_base, stream_add):
    check(lib().bem_bnn_bank_sample_f32(_p(bank.segs), _p(bank.blks), bank.nblk, _p(bank.pm), _p(bank.pr), _p(bank.w), _p(bank.eps), _p(bank.gw),
                                        float(decay), _p(decay_dev), seed, stream_base, _p(stream_add), _stream()), "bnn_bank_sample")


def bnn_ebank_sample(bank, seed, stream_base):
    check(lib().bem_bnn_ebank_sample_f32(_p(bank.segs), _p(bank.blks), bank.nblk, _p(bank.arena), seed, stream_base, _stream()), "bnn_ebank_sample")


def pack_pw_weight_jobs(jobs, blks, nblk, arena):
    check(lib().bem_pack_pw_weight_x6_jobs(_p(jobs), _p(blks), nblk, _p(arena), _stream()), "pack_pw_weight_x6_jobs")


def bnn_bank_kl_(bank, out):
    _chk(out, "out")
    check(lib().bem_bnn_bank_kl_f32(_p(bank.segs), _p(bank.blks), bank.nblk, _p(bank.pm), _p(bank.pr), _p(out), _stream()), "bnn_bank_kl")


def bnn_bank_kl_bwd_(bank, g):
    _chk(g, "g")
    check(lib().bem_bnn_bank_kl_bwd_f32(_p(bank.segs), _p(bank.blks), bank.nblk, _p(bank.pm), _p(bank.pr), _p(g), _stream()), "bnn_bank_kl_bwd")


def bnn_bank_reparam_bwd_(bank):
    check(lib().bem_bnn_bank_reparam_bwd_f32(_p(bank.segs), _p(bank.blks), bank.nblk, _p(bank.gw), _p(bank.eps), _stream()), "bnn_bank_reparam_bwd")


def bnn_kl_(mu, rho, prior_mu, prior_rho, out):
    """out[0] += kl_div(mu, softplus(rho), prior_mu, softplus(prior_rho)).mean()"""
    for t, nm in ((prior_mu, "prior_mu"), (prior_rho, "prior_rho"), (mu, "mu"), (rho, "rho"), (out, "out")):
        _chk(t, nm)
    if not (prior_mu.shape == prior_rho.shape == mu.shape == rho.shape) or out.numel() != 1:
        raise ValueError("bnn_kl_: shapes")
    check(lib().bem_bnn_kl_f32(_p(mu), _p(rho), _p(prior_mu), _p(prior_rho), mu.numel(), _p(out), _stream()), "bnn_kl")


def bnn_kl_bwd_(mu, rho, prior_mu, prior_rho, g, dmu, drho):
    for t, nm in ((prior_mu, "prior_mu"), (prior_rho, "prior_rho"), (mu, "mu"), (rho, "rho"), (g, "g"), (dmu, "dmu"), (drho, "drho")):
        _chk(t, nm)
    if not (prior_mu.shape == prior_rho.shape == mu.shape == rho.shape == dmu.shape == drho.shape) or g.numel() != 1:
        raise ValueError("bnn_kl_bwd_: shapes")
    check(lib().bem_bnn_kl_bwd_f32(_p(mu), _p(rho), _p(prior_mu), _p(prior_rho), mu.numel(), _p(g), _p(dmu), _p(drho), _stream()), "bnn_kl_bwd")


def bnn_reparam_bwd_(gw, eps, rho, dmu, drho):
    for t, nm in ((gw, "gw"), (eps, "eps"), (rho, "rho"), (dmu, "dmu"), (drho, "drho")):
        _chk(t, nm)
    if not (gw.numel() == eps.numel() == rho.numel() == dmu.numel() == drho.numel()):
        raise ValueError("bnn_reparam_bwd_: sizes differ")
    check(lib().bem_bnn_reparam_bwd_f32(_p(gw), _p(eps), _p(rho), _p(dmu), _p(drho), rho.numel(), _stream()), "bnn_reparam_bwd")


def mask_token(fea, mask, token):
    _chk(fea, "fea"); _chk(mask, "mask"); _chk(token, "token")
    B, C, H, W = fea.shape
    if tuple(mask.shape) != (B, H, W) or token.numel() != C:
        raise ValueError(f"mask_token: mask {tuple(mask.shape)} / token {tuple(token.shape)} vs features {tuple(fea.shape)}")
    out = torch.empty_like(fea)
    check(lib().bem_mask_token_f32(_p(fea), _p(mask), _p(token), _p(out), B, C, H, W, _stream()), "mask_token")
    return out


def mask_token_bwd(dout, mask, dtoken):
    _chk(dout, "dout"); _chk(mask, "mask"); _chk(dtoken, "dtoken")
    B, C, H, W = dout.shape
    if tuple(mask.shape) != (B, H, W) or dtoken.numel() != C:
        raise ValueError("mask_token_bwd: shapes")
    dfea = torch.empty_like(dout)
    check(lib().bem_mask_token_bwd_f32(_p(dout), _p(mask), _p(dfea), _p(dtoken), B, C, H, W, _stream()), "mask_token_bwd")
    return dfea


def depth_to_space(d4):
    _chk(d4, "d4")
    B, C4, h, w = d4.shape
    if C4 % 4:
        raise ValueError("depth_to_space: channels % 4")
    dx = torch.empty(B, C4 // 4, 2 * h, 2 * w, device=d4.device, dtype=d4.dtype)
    check(lib().bem_depth_to_space_f32(_p(d4), _p(dx), B, C4 // 4, 2 * h, 2 * w, _stream()), "depth_to_space")
    return dx


def prelu(x, slope):
    _chk(x, "x"); _chk(slope, "slope")
    if slope.numel() != 1:
        raise ValueError("prelu: one shared slope (nn.PReLU())")
    out = torch.empty_like(x)
    check(lib().bem_prelu_f32(_p(x), _p(slope), _p(out), x.numel(), _stream()), "prelu")
    return out


def prelu_bwd(x, slope, dout, dslope):
    _chk(x, "x"); _chk(slope, "slope"); _chk(dout, "dout"); _chk(dslope, "dslope")
    if slope.numel() != 1 or dslope.numel() != 1 or dout.shape != x.shape:
        raise ValueError("prelu_bwd: shapes")
    dx = torch.empty_like(x)
    check(lib().bem_prelu_bwd_f32(_p(x), _p(slope), _p(dout), _p(dx), _p(dslope), x.numel(), _stream()), "prelu_bwd")
    return dx


def bilinear_up_bwd(dout, s):
    _chk(dout, "dout")
    B, C, Ho, Wo = dout.shape
    if Ho % s or Wo % s:
        raise ValueError("bilinear_up_bwd: output size not a multiple of the scale")
    dx = torch.empty(B, C, Ho // s, Wo // s, device=dout.device, dtype=dout.dtype)
    check(lib().bem_bilinear_up_bwd_f32(_p(dout), _p(dx), B, C, Ho // s, Wo // s, int(s), _stream()), "bilinear_up_bwd")
    return dx


# --------------------------------------------------------------------------- Bayesian / MC ----
def bnn_sample(mu, rho, nsets, eps=None, seed=0, stream_id=0, stream_add=None):
    """w[s] = mu + log1p(exp(rho)) * eps[s];  eps None -> Philox N(0,1) keyed by (seed, stream_id [+ stream_add[0], a one-element int64
    device tensor read by the kernel: the per-iteration part of the id of a graph-captured step])."""
    _chk(mu, "mu"); _chk(rho, "rho"); _chk(eps, "eps", optional=True)
    n = mu.numel()
    if rho.numel() != n or (eps is not None and eps.numel() != nsets * n):
        raise ValueError("bnn_sample: shapes")
    out = torch.empty((nsets,) + tuple(mu.shape), device=mu.device, dtype=mu.dtype)
    check(lib().bem_bnn_sample_f32(_p(mu), _p(rho), _p(eps), _p(out), nsets, n, seed, stream_id, _p(stream_add), _stream()), "bnn_sample")
    return out


def bnn_sample_packed(mu, rho, nsets, M, K, eps=None, seed=0, stream_id=0, sigma_given=False, stream_add=None):
    """bnn_sample + pack_pw_weight(x6) in one kernel: (nsets, packed(M, K)) GEMM weights of a Bayesian 1x1 layer.
    sigma_given: ``rho`` already holds sigma = log1p(exp(rho)) (it does not depend on the sample)."""
    _chk(mu, "mu"); _chk(rho, "rho"); _chk(eps, "eps", optional=True)
    if mu.numel() != M * K or rho.numel() != M * K or (eps is not None and eps.numel() != nsets * M * K):
        raise ValueError("bnn_sample_packed: shapes")
    out = torch.empty(nsets, packed_elems(M, K, True), device=mu.device, dtype=mu.dtype)
    check(lib().bem_bnn_sample_pack_x6(_p(mu), _p(rho), _p(eps), _p(out), nsets, M, K, seed, stream_id, _p(stream_add), int(bool(sigma_given)), _stream()), "bnn_sample_pack_x6")
    out._bem_mk = (M, K)
    return out


def pad_reflect(x, Hp, Wp):
    """(B,C,H,W) -> (B,C,Hp,Wp), reflect-padded at the bottom / right (numpy 'reflect')."""
    _chk(x, "x")
    B, C, H, W = x.shape
    if Hp == H and Wp == W:
        return x
    out = torch.empty(B, C, Hp, Wp, device=x.device, dtype=x.dtype)
    check(lib().bem_pad_reflect_f32(_p(x), _p(out), B * C, H, W, Hp, Wp, _stream()), "pad_reflect")
    return out


def resize_down(x, s):
    """cv2.resize(fx=fy=1/s, INTER_LINEAR) of (B,C,Hp,Wp) planes for even s dividing Hp, Wp."""
    _chk(x, "x")
    B, C, Hp, Wp = x.shape
    out = torch.empty(B, C, Hp // s, Wp // s, device=x.device, dtype=x.dtype)
    check(lib().bem_resize_down_f32(_p(x), _p(out), B * C, Hp, Wp, s, _stream()), "resize_down")
    return out


def randn(shape, device, seed=0, stream_id=0, stream_add=None):
    out = torch.empty(shape, device=device, dtype=torch.float32)
    check(lib().bem_randn_f32(_p(out), out.numel(), seed, stream_id, _p(stream_add), _stream()), "randn")
    return out


def plane_mean(x, h=None, w=None):
    """Mean over the top-left (h,w) window of every (b,c) plane -> (B,C)."""
    _chk(x, "x")
    B, C, Hs, Ws = x.shape
    h, w = h or Hs, w or Ws
    out = torch.empty(B, C, device=x.device, dtype=x.dtype)
    check(lib().bem_plane_mean_f32(_p(x), _p(out), B * C, Hs, Ws, h, w, _stream()), "plane_mean")
    return out


# --------------------------------------------------------------------------- DecompDualBranch's bottleneck blocks ----
def row_scale(w, scale):
    """w (M, ...) * scale (M) along the first axis."""
    _chk(w, "w"); _chk(scale, "scale")
    M = w.shape[0]
    if scale.numel() != M:
        raise ValueError("row_scale: one factor per row")
    out = torch.empty_like(w)
    check(lib().bem_row_scale_f32(_p(w), _p(scale), _p(out), M, w.numel() // M, _stream()), "row_scale")
    return out


def se_gate(x, w1, w2, want_mean=False):
    """SEBlock's per-channel gate (B,C) = sigmoid(w2 relu(w1 mean_hw(x)))."""
    _chk(x, "x"); _chk(w1, "w1"); _chk(w2, "w2")
    B, C = x.shape[0], x.shape[1]
    Cr = w1.shape[0]
    if tuple(w1.shape) != (Cr, C) or tuple(w2.shape) != (C, Cr):
        raise ValueError("se_gate: weight shapes")
    mean = plane_mean(x)
    y = torch.empty(B, C, device=x.device, dtype=x.dtype)
    check(lib().bem_se_gate_f32(_p(mean), _p(w1), _p(w2), _p(y), B, C, Cr, _stream()), "se_gate")
    return (y, mean) if want_mean else y


def spatial_attention(x, w, chan_scale=None, want_map=False):
    """x * chan_scale * sigmoid(conv_kxk([mean_c, max_c](x * chan_scale))); w (1,2,k,k).  want_map: also the (B,2,H,W) map (for the backward)."""
    _chk(x, "x"); _chk(w, "w"); _chk(chan_scale, "chan_scale", optional=True)
    B, C, H, W = x.shape
    k = w.shape[-1]
    if tuple(w.shape) != (1, 2, k, k) or (chan_scale is not None and tuple(chan_scale.shape) != (B, C)):
        raise ValueError("spatial_attention: weight / gate shapes")
    ws = torch.empty(B, 2, H, W, device=x.device, dtype=x.dtype)
    out = torch.empty_like(x)
    check(lib().bem_spatial_attention_f32(_p(x), _p(chan_scale), _p(w), _p(ws), _p(out), B, C, H, W, k, _stream()), "spatial_attention")
    return (out, ws) if want_map else out


def spatial_attention_bwd(x, dout, amap, w, dw):
    """Backward of spatial_attention(x, w): returns dx; dw (1,2,k,k) accumulated."""
    for t, n in ((x, "x"), (dout, "dout"), (amap, "map"), (w, "w"), (dw, "dw")):
        _chk(t, n)
    B, C, H, W = x.shape
    dpre = torch.empty(B, H, W, device=x.device, dtype=x.dtype)
    dx = torch.empty_like(x)
    check(lib().bem_spatial_attention_bwd_f32(_p(x), _p(dout), _p(amap), _p(w), _p(dpre), _p(dx), _p(dw), B, C, H, W, w.shape[-1], _stream()),
          "spatial_attention_bwd")
    return dx


def chan_scale(x, scale, add=None, add_bc=None, add_bc_scale=1.0):
    """scale[(b,) c] * x (+ add) (+ add_bc[b, c] * add_bc_scale); scale (C) or (B, C)."""
    _chk(x, "x"); _chk(scale, "scale"); _chk(add, "add", optional=True); _chk(add_bc, "add_bc", optional=True)
    B, C = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    if scale.numel() not in (C, B * C) or (add is not None and add.shape != x.shape) or (add_bc is not None and add_bc.numel() != B * C):
        raise ValueError("chan_scale: shapes")
    out = torch.empty_like(x)
    check(lib().bem_chan_scale_f32(_p(x), _p(scale), C if (scale.numel() == B * C and scale.dim() > 1) else 0, _p(add), _p(add_bc), float(add_bc_scale), _p(out),
                                   B, C, HW, _stream()), "chan_scale")
    return out


def chan_dot(a, b, out=None):
    """out is None: (B,C) = sum_p a b per image; out (C): += sum over images and pixels (a parameter's gradient)."""
    _chk(a, "a"); _chk(b, "b"); _chk(out, "out", optional=True)
    B, C = a.shape[0], a.shape[1]
    HW = a[0, 0].numel()
    if b.shape != a.shape or (out is not None and out.numel() != C):
        raise ValueError("chan_dot: shapes")
    per = out is None
    if per:
        out = torch.empty(B, C, device=a.device, dtype=a.dtype)
    check(lib().bem_chan_dot_f32(_p(a), _p(b), _p(out), B, C, HW, int(per), _stream()), "chan_dot")
    return out


def se_gate_bwd(mean, w1, w2, y, dy, dw1, dw2):
    for t, n in ((mean, "mean"), (w1, "w1"), (w2, "w2"), (y, "y"), (dy, "dy"), (dw1, "dw1"), (dw2, "dw2")):
        _chk(t, n)
    B, C = mean.shape
    dmean = torch.empty_like(mean)
    check(lib().bem_se_gate_bwd_f32(_p(mean), _p(w1), _p(w2), _p(y), _p(dy), _p(dmean), _p(dw1), _p(dw2), B, C, w1.shape[0], _stream()), "se_gate_bwd")
    return dmean


def cond_postproc(pred, target_mean, noise, samples_per_image, noise_level):
    _chk(pred, "pred"); _chk(target_mean, "target_mean", optional=True); _chk(noise, "noise", optional=True)
    Bn, C, h, w = pred.shape
    if C != 3 or Bn % samples_per_image:
        raise ValueError("cond_postproc: expects (Bn,3,h,w)")
    if target_mean is not None and tuple(target_mean.shape) != (Bn // samples_per_image, 3):
        raise ValueError("cond_postproc: target_mean shape")
    if noise is not None and noise.shape != pred.shape:
        raise ValueError("cond_postproc: noise shape")
    out = torch.empty_like(pred)
    check(lib().bem_cond_postproc_f32(_p(pred), _p(target_mean), _p(noise), _p(out), Bn, h, w, samples_per_image,
                                      float(noise_level), _stream()), "cond_postproc")
    return out


def candidate_finalize(pred, target, samples_per_image, h, w, gt_mean):
    """pred (Bn,3,Hp,Wp), target (n_img,3,h,w)|None -> (final (Bn,3,h,w), psnr (Bn))."""
    _chk(pred, "pred"); _chk(target, "target", optional=True)
    Bn, C, Hp, Wp = pred.shape
    if C != 3 or Bn % samples_per_image or h > Hp or w > Wp:
        raise ValueError("candidate_finalize: shapes")
    if target is not None and tuple(target.shape) != (Bn // samples_per_image, 3, h, w):
        raise ValueError("candidate_finalize: target shape")
    fin = torch.empty(Bn, 3, h, w, device=pred.device, dtype=pred.dtype)
    ps = torch.zeros(Bn, device=pred.device, dtype=pred.dtype)
    ws = torch.empty(7 * Bn, device=pred.device, dtype=torch.float64)
    check(lib().bem_candidate_finalize_f32(_p(pred), _p(target), _p(fin), _p(ps), _p(ws), Bn, samples_per_image, Hp, Wp, h, w,
                                           int(bool(gt_mean)), _stream()), "candidate_finalize")
    return fin, ps


def select_best(final, psnr, samples_per_image):
    """eval.py:284-285 on the device: (best (B) int32, best_psnr (B), best_images (B,3,h,w)); row = image*N + sample."""
    _chk(final, "final"); _chk(psnr, "psnr")
    Bn = final.shape[0]
    N = samples_per_image
    if Bn % N or psnr.numel() != Bn:
        raise ValueError("select_best: shapes")
    B = Bn // N
    best = torch.empty(B, device=final.device, dtype=torch.int32)
    bp = torch.empty(B, device=final.device, dtype=torch.float32)
    img = torch.empty((B,) + tuple(final.shape[1:]), device=final.device, dtype=final.dtype)
    check(lib().bem_select_best_f32(_p(final), _p(psnr), _p(best), _p(bp), _p(img), B, N, final[0].numel(), _stream()), "select_best")
    return best, bp, img


def ssim(final, target, samples_per_image):
    """Enhancement/utils.py calculate_ssim per candidate: final (Bn,3,h,w) in [0,1], target (Bn/N,3,h,w) -> (Bn) f32."""
    _chk(final, "final"); _chk(target, "target")
    Bn, C, h, w = final.shape
    if C != 3 or Bn % samples_per_image or tuple(target.shape) != (Bn // samples_per_image, 3, h, w) or h <= 10 or w <= 10:
        raise ValueError("ssim: shapes (3-channel images larger than the 11x11 window)")
    out = torch.empty(Bn, device=final.device, dtype=torch.float32)
    ws = torch.empty(Bn, device=final.device, dtype=torch.float64)
    check(lib().bem_ssim_f32(_p(final), _p(target), _p(out), _p(ws), Bn, samples_per_image, h, w, _stream()), "ssim")
    return out


def select_scores(final, s1, samples_per_image, s2=None, weight=1.0, rule="weighted"):
    """Per-image selection on the device (eval.py:268-297): rule 'weighted' (weight s1/max + (1-weight) s2/max), 'max', 'min';
    first index on ties.  Returns (best (B) int32, best s1 (B), best s2 (B)|None, best images (B,3,h,w)|None)."""
    _chk(s1, "s1"); _chk(s2, "s2", optional=True); _chk(final, "final", optional=True)
    N = samples_per_image
    Bn = s1.numel()
    if Bn % N or (s2 is not None and s2.numel() != Bn) or (final is not None and final.shape[0] != Bn):
        raise ValueError("select_scores: shapes")
    r = {"weighted": 0, "max": 1, "min": 2}[rule]
    B = Bn // N
    best = torch.empty(B, device=s1.device, dtype=torch.int32)
    b1 = torch.empty(B, device=s1.device, dtype=torch.float32)
    b2 = torch.empty(B, device=s1.device, dtype=torch.float32) if s2 is not None else None
    img = torch.empty((B,) + tuple(final.shape[1:]), device=s1.device, dtype=final.dtype) if final is not None else None
    check(lib().bem_select_scores_f32(_p(final), _p(s1), _p(s2), float(weight), r, _p(best), _p(b1), _p(b2), _p(img), B, N,
                                      (final[0].numel() if final is not None else 0), _stream()), "select_scores")
    return best, b1, b2, img


def mc_mean(raw, target, samples_per_image, h, w, gt_mean):
    """Monte-Carlo mean prediction (eval.py:224-225,308-314): raw (Bn,3,Hp,Wp) -> (B,3,h,w)."""
    _chk(raw, "raw"); _chk(target, "target", optional=True)
    Bn, C, Hp, Wp = raw.shape
    N = samples_per_image
    if C != 3 or Bn % N or h > Hp or w > Wp or (gt_mean and (target is None or tuple(target.shape) != (Bn // N, 3, h, w))):
        raise ValueError("mc_mean: shapes")
    B = Bn // N
    out = torch.empty(B, 3, h, w, device=raw.device, dtype=raw.dtype)
    ws = torch.empty(2 * B, device=raw.device, dtype=torch.float64) if gt_mean else None
    check(lib().bem_mc_mean_f32(_p(raw), _p(target if gt_mean else None), _p(out), _p(ws), B, N, Hp, Wp, h, w, int(bool(gt_mean)), _stream()), "mc_mean")
    return out


# --------------------------------------------------------------------------- training step ----
# Backward kernels + optimizer.  Parameter-gradient outputs (dw, dbias, dgamma, ...) are
# ACCUMULATED INTO: they are views of the parameters' .grad buffers.
def l1_loss(pred, gt, weight=1.0):
    """loss (1,) = weight * mean |pred - gt|."""
    _chk(pred, "pred"); _chk(gt, "gt")
    if pred.shape != gt.shape:
        raise ValueError("l1_loss: shapes differ")
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    ws = torch.empty(1, device=pred.device, dtype=torch.float64)
    check(lib().bem_l1_loss_f32(_p(pred), _p(gt), _p(None), _p(loss), _p(ws), pred.numel(), float(weight), _p(None), _stream()), "l1_loss")
    return loss


def l1_loss_bwd(pred, gt, weight=1.0, gmul=None):
    """dpred = gmul * weight * sign(pred - gt) / numel  (gmul: device scalar dL/dloss, None = 1)."""
    _chk(pred, "pred"); _chk(gt, "gt"); _chk(gmul, "gmul", optional=True)
    dp = torch.empty_like(pred)
    ws = torch.empty(1, device=pred.device, dtype=torch.float64)
    check(lib().bem_l1_loss_f32(_p(pred), _p(gt), _p(dp), _p(None), _p(ws), pred.numel(), float(weight), _p(gmul), _stream()), "l1_loss_bwd")
    return dp


def iwt_hamilton_bwd(q1w, q2w, dout):
    _chk(q1w, "q1w"); _chk(q2w, "q2w"); _chk(dout, "dout")
    B, C, h, w = q1w.shape
    if C != 16 or q2w.shape != q1w.shape or tuple(dout.shape) != (B, 3, 2 * h, 2 * w):
        raise ValueError("iwt_hamilton_bwd: shapes")
    d1, d2 = torch.empty_like(q1w), torch.empty_like(q2w)
    check(lib().bem_iwt_hamilton_bwd_f32(_p(q1w), _p(q2w), _p(dout), _p(d1), _p(d2), B, h, w, _stream()), "iwt_hamilton_bwd")
    return d1, d2


def pixel_unshuffle2(x):
    """nn.PixelUnshuffle(2): (B,C,2H,2W) -> (B,4C,H,W)."""
    _chk(x, "x")
    B, C, H2, W2 = x.shape
    if H2 % 2 or W2 % 2:
        raise ValueError("pixel_unshuffle2: even H, W required")
    out = torch.empty(B, 4 * C, H2 // 2, W2 // 2, device=x.device, dtype=x.dtype)
    check(lib().bem_pixel_unshuffle2_f32(_p(x), _p(out), B, C, H2 // 2, W2 // 2, _stream()), "pixel_unshuffle2")
    return out


def channel_sum_(x, out):
    """out[c] += sum_{b,p} x[b][c][p]."""
    _chk(x, "x"); _chk(out, "out")
    B, C = x.shape[0], x.shape[1]
    if out.numel() != C:
        raise ValueError("channel_sum: out size")
    check(lib().bem_channel_sum_f32(_p(x), _p(out), B, C, x[0, 0].numel(), _stream()), "channel_sum")
    return out


def add(a, b, alpha=1.0):
    """a + alpha * b."""
    _chk(a, "a"); _chk(b, "b")
    if a.shape != b.shape:
        raise ValueError("add: shapes differ")
    out = torch.empty_like(a)
    check(lib().bem_add_f32(_p(a), _p(b), _p(out), a.numel(), float(alpha), _stream()), "add")
    return out


def ln_bwd(x1, dn, gamma, beta, eps, dgamma, dbeta, x2=None, dres=None, want_n=True):
    """LayerNorm2d backward over channels of x = x1 (+ x2): returns (dx, n = LN(x) | None); dgamma / dbeta accumulated."""
    for n, t in (("x1", x1), ("dn", dn), ("gamma", gamma), ("beta", beta), ("dgamma", dgamma), ("dbeta", dbeta)):
        _chk(t, n)
    _chk(x2, "x2", optional=True); _chk(dres, "dres", optional=True)
    B, C = x1.shape[0], x1.shape[1]
    if dn.shape != x1.shape or (x2 is not None and x2.shape != x1.shape) or (dres is not None and dres.shape != x1.shape):
        raise ValueError("ln_bwd: activation shapes")
    if gamma.numel() != C or beta.numel() != C or dgamma.numel() != C or dbeta.numel() != C:
        raise ValueError("ln_bwd: parameter sizes")
    dx = torch.empty_like(x1)
    nn_ = torch.empty_like(x1) if want_n else None
    check(lib().bem_ln_bwd_f32(_p(x1), _p(x2), _p(dn), _p(gamma), _p(beta), float(eps), _p(dres), _p(dx), _p(nn_), _p(dgamma), _p(dbeta),
                               B, C, x1[0, 0].numel(), _stream()), "ln_bwd")
    return dx, nn_


def ln_fwd(x1, gamma, beta, eps, x2=None):
    _chk(x1, "x1"); _chk(gamma, "gamma"); _chk(beta, "beta"); _chk(x2, "x2", optional=True)
    B, C = x1.shape[0], x1.shape[1]
    if gamma.numel() != C or beta.numel() != C or (x2 is not None and x2.shape != x1.shape):
        raise ValueError("ln_fwd: shapes")
    out = torch.empty_like(x1)
    check(lib().bem_ln_fwd_f32(_p(x1), _p(x2), _p(gamma), _p(beta), float(eps), _p(out), B, C, x1[0, 0].numel(), _stream()), "ln_fwd")
    return out


def dwact_bwd(t, w, bias, dout, dw, dbias, mode):
    """Backward of dwconv3x3(mode) through the activation: returns dpre (like t); dw (Cw,1,3,3) / dbias accumulated."""
    _chk(t, "t"); _chk(w, "w"); _chk(bias, "bias", optional=True); _chk(dout, "dout"); _chk(dw, "dw"); _chk(dbias, "dbias", optional=True)
    B, Cw, H, W = t.shape
    Cout = Cw // 2 if mode == 2 else Cw
    if tuple(dout.shape) != (B, Cout, H, W) or w.numel() != Cw * 9 or dw.numel() != Cw * 9 or (bias is None) != (dbias is None):
        raise ValueError("dwact_bwd: shapes")
    if bias is not None and (bias.numel() != Cw or dbias.numel() != Cw):
        raise ValueError("dwact_bwd: bias shapes")
    dpre = torch.empty_like(t)
    check(lib().bem_dwact_bwd_f32(_p(t), _p(w), _p(bias), _p(dout), _p(dpre), _p(dw), _p(dbias), B, Cout, H, W, mode, _stream()), "dwact_bwd")
    return dpre


WGRAD_X6 = os.environ.get("BEM_WGRAD_X6", "1") != "0"      # 1x1 weight gradients on the bf16 matrix cores (no LDS transposes) when L % 32 == 0
# below this many pixels per launch the x6 form's fixed costs (one wave per SIMD, second reduction launch) lose to the f32-MFMA kernel:
# Stage-I training (8 x 8 planes) 34.7 -> 23.8 ms per step
WGRAD_X6_MIN_PIXELS = int(os.environ.get("BEM_WGRAD_X6_MIN_PIXELS", "16384"))
_WGX_WS = {}                                               # per-device scratch of the x6 weight-gradient kernel (stream-ordered reuse)


def pw_wgrad_(dy, x1, dw, x2=None, dbias=None, blk_rows=0, perm=(0, 1, 2, 3), dy_bstride=0, M=None):
    """dw (M, C1 + C2) += dy . cat(x1, x2)^T over batch and pixels; dbias (M) += row sums of dy.
    dy may be a channel slice of a wider tensor (pass M and dy_bstride)."""
    _chk(x1, "x1"); _chk(x2, "x2", optional=True); _chk(dw, "dw"); _chk(dbias, "dbias", optional=True)
    if not dy.is_cuda or dy.dtype != torch.float32:
        raise native.BemNativeError("pw_wgrad: dy must be a float32 CUDA/HIP tensor")
    B, C1 = x1.shape[0], x1.shape[1]
    L = x1[0, 0].numel()
    C2 = 0 if x2 is None else x2.shape[1]
    if M is None:
        _chk(dy, "dy")
        M = dy.shape[1]
    if dy.shape[0] != B or (x2 is not None and (x2.shape[0] != B or x2[0, 0].numel() != L)):
        raise ValueError("pw_wgrad: batch / pixel counts")
    if dw.numel() != M * (C1 + C2) or (dbias is not None and dbias.numel() != M):
        raise ValueError(f"pw_wgrad: dw has {dw.numel()} elements, expected {M} x {C1 + C2}")
    a = native.WgradArgs()
    a.dy, a.dy_bstride, a.M = dy.data_ptr(), dy_bstride, M
    a.x1, a.x1_bstride, a.C1 = x1.data_ptr(), 0, C1
    a.x2, a.x2_bstride, a.C2 = (x2.data_ptr() if x2 is not None else 0), 0, C2
    a.dw, a.ldw, a.blk_rows = dw.data_ptr(), C1 + C2, blk_rows
    for i in range(4):
        a.perm[i] = perm[i]
    a.dbias = dbias.data_ptr() if dbias is not None else 0
    a.B, a.L = B, L
    if WGRAD_X6 and USE_X6 and L % 32 == 0 and B * L >= WGRAD_X6_MIN_PIXELS:
        n = lib().bem_pw_wgrad_x6_ws_elems(M, C1 + C2, B, L)
        ws = _WGX_WS.get(dw.device)
        if ws is None or ws.numel() < n:
            ws = _WGX_WS[dw.device] = torch.empty(max(n, 1 << 20), device=dw.device, dtype=torch.float32)
        check(lib().bem_pw_wgrad_x6_f32(ctypes.byref(a), _p(ws), ws.numel(), _stream()), "pw_wgrad_x6")
    else:
        check(lib().bem_pw_wgrad_f32(ctypes.byref(a), _stream()), "pw_wgrad")
    return dw


def conv_wgrad_(dy, x, dw, dbias=None, stride=1, pad=1, cin_slice=None):
    """dw (Cout,Cin,KH,KW) += weight gradient of conv2d(x, w, stride, pad); dbias (Cout) += sum dy."""
    _chk(dy, "dy"); _chk(x, "x"); _chk(dw, "dw"); _chk(dbias, "dbias", optional=True)
    B, Ct, H, W = x.shape
    Cout, Cin, KH, KW = dw.shape
    c0 = 0
    if cin_slice is not None:
        c0, cs = cin_slice
        if cs != Cin or c0 + Cin > Ct:
            raise ValueError("conv_wgrad: channel slice")
    elif Ct != Cin:
        raise ValueError("conv_wgrad: channels")
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    if tuple(dy.shape) != (B, Cout, Ho, Wo) or (dbias is not None and dbias.numel() != Cout):
        raise ValueError("conv_wgrad: dy / dbias shapes")
    check(lib().bem_conv_wgrad_f32(_p(dy), ctypes.c_void_p(x.data_ptr() + 4 * c0 * H * W), Ct * H * W, _p(dw), _p(dbias), B, Cin, H, W, Cout,
                                   KH, KW, stride, pad, _stream()), "conv_wgrad")
    return dw


def ss2d_scan_bwd(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dAlog, dDs, ddtw, ddtb):
    """Backward of ss2d_scan: returns (dx0, dx1, dxd0, dxd1); parameter gradients accumulated into dAlog (4C), dDs (4C),
    ddtw (4,C,R), ddtb (4,C)."""
    for n, t in (("x0", x0), ("x1", x1), ("dy0", dy0), ("dy1", dy1), ("dtw", dtw), ("dtb", dtb), ("A", A), ("Ds", Ds), ("dAlog", dAlog),
                 ("dDs", dDs), ("ddtw", ddtw), ("ddtb", ddtb)):
        _chk(t, n)
    B, C, L = x0.shape
    R = dtw.shape[2]
    if x1.shape != x0.shape or dy0.shape != x0.shape or dy1.shape != x0.shape or xd0.shape != (B, 2, R + 2, L) or xd1.shape != xd0.shape:
        raise ValueError("ss2d_scan_bwd: activation shapes")
    if dtw.shape != (4, C, R) or dtb.shape != (4, C) or A.numel() != 4 * C or Ds.numel() != 4 * C or dAlog.numel() != 4 * C or dDs.numel() != 4 * C \
            or ddtw.numel() != 4 * C * R or ddtb.numel() != 4 * C:
        raise ValueError("ss2d_scan_bwd: parameter shapes")
    bs = []
    for n, t in (("xd0", xd0), ("xd1", xd1)):
        if not t.is_cuda or t.dtype != torch.float32:
            raise native.BemNativeError(f"{n} must be a float32 CUDA/HIP tensor")
        if t.stride()[1:] != ((R + 2) * L, L, 1) or (B > 1 and (t.stride(0) < 2 * (R + 2) * L or (L % 4 == 0 and t.stride(0) % 4))):
            raise ValueError(f"{n}: only the batch stride may differ from a contiguous (B,2,R+2,L) tensor")
        bs.append(t.stride(0) if B > 1 else 0)
    dx0, dx1 = torch.empty_like(x0), torch.empty_like(x0)
    dxd0 = torch.empty(B, 2, R + 2, L, device=x0.device, dtype=x0.dtype)
    dxd1 = torch.empty(B, 2, R + 2, L, device=x0.device, dtype=x0.dtype)
    check(lib().bem_ss2d_scan_bwd_f32(_p(x0), _p(x1), _p(xd0), _p(xd1), _p(dy0), _p(dy1), _p(dtw), _p(dtb), _p(A), _p(Ds), _p(dx0), _p(dx1),
                                      _p(dxd0), _p(dxd1), _p(dAlog), _p(dDs), _p(ddtw), _p(ddtb), B, C, L, R, bs[0], bs[1], _stream()),
          "ss2d_scan_bwd")
    return dx0, dx1, dxd0, dxd1


def grad_sumsq(g, acc):
    """acc (1,) f64 = sum g^2 over the flat gradient buffer."""
    _chk(g, "g"); _chk(acc, "acc", dtype=torch.float64)
    check(lib().bem_grad_sumsq_f32(_p(g), g.numel(), _p(acc), _stream()), "grad_sumsq")
    return acc


def adamw_step_(p, g, m, v, lr, betas, eps, weight_decay, step, max_norm=0.0, sumsq=None, norm_out=None, hyper=None):
    """hyper: optional device tensor [lr, 1 - beta1^t, sqrt(1 - beta2^t)] read by the kernel in place of ``lr`` / ``step``."""
    _chk(hyper, "hyper", optional=True)
    for n, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, n)
    _chk(sumsq, "sumsq", dtype=torch.float64, optional=True); _chk(norm_out, "norm_out", optional=True)
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n:
        raise ValueError("adamw_step: buffer sizes differ")
    check(lib().bem_adamw_step_f32(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                   int(step), float(max_norm), _p(sumsq), _p(norm_out), _p(hyper), _stream()), "adamw_step")


# --------------------------------------------------------------------------- launch timing ----
# bench.py asks for ONE op's launches to be bracketed by HIP events on the launch stream (torch's
# current stream is the stream every wrapper launches on), together with that launch's algorithmic
# bytes / flops.  Disabled (zero overhead beyond a None check) outside bench.py.
_PROF = None
# profile key -> (op wrapper it belongs to, roofline bound, kernel symbol reported to bench.py, predicate on the call)
_KEYS = {
    # the register-resident x6 GEMM for K <= 48 (level-0 in_proj / project_in / x_proj and the small Stage-I layers):
    # every launch of the (vectorised, single-input or concat) instance, whatever M and L
    "pw_x6_res<3,2,1>": ("pw_gemm", "hbm", "pw_x6_res_kernel<3, 2, 1, false, true>",
                         lambda K, M, ln, L, mode: USE_X6 and K <= 48 and L % 2 == 0 and mode != 1),
    # the streaming x6 GEMM with two M-tiles per pass (project_out / out_proj / fuse 1x1 at K > 48 without LayerNorm):
    # the kernel with the largest share of the step in profiles/r01_bench_kernel_stats.csv
    "pw_x6_stream<2>": ("pw_gemm", "hbm", "pw_x6_stream_kernel<2, 2, false, true, false>",
                        lambda K, M, ln, L, mode: USE_X6 and K > 48 and not ln and M > 32 and L % 2 == 0 and mode != 1),
    "pw_gemm": ("pw_gemm", "mfma", "pw_gemm* (all variants)", lambda K, M, ln, L, mode: True),
    # the whole gdMlp branch in one kernel: x in, out out -- 8 bytes per element of x are its algorithmic bytes
    # (HBM: 42 us at level 0) -- but its two GEMMs (2Hd x C and C x Hd per pixel) evaluated as six bf16 limb products are 242 GFLOP on the
    # matrix cores (97 us at the dense bf16 peak): the matrix pipe is the roofline that bounds it.  flops = 6 x the f32 GEMM flops
    # (what the x6 scheme must issue for the output pixels; halo and padding MFMAs are waste, not work).
    "gdmlp_x6<3>": ("gdmlp_x6", "mfma_bf16", "gdmlp_x6_kernel<3, 2, 2, false>", lambda C: 32 < C <= 48),
    "gdmlp_x6<5>": ("gdmlp_x6", "mfma_bf16", "gdmlp_x6_kernel<5, 3, 1, false>", lambda C: 64 < C <= 80),
    "conv2d": ("conv2d", "mfma", "conv2d_kernel", None),
    "dwconv3x3": ("dwconv3x3", "hbm", "dwconv3x3_kernel", None),
    "ss2d_scan": ("ss2d_scan", "hbm", "ss2d_scan_kernel", None),
    "transpose_planes": ("transpose_planes", "hbm", "transpose_planes_kernel", None),
    # training step (bench.py --config train)
    "pw_wgrad": ("pw_wgrad_", "hbm", "wgrad_x6_kernel<2, 2> + wgrad_x6_reduce_kernel (1x1 weight gradients; wgrad_kernel<*> for L % 32 != 0)", None),
    "conv_wgrad": ("conv_wgrad_", "mfma", "wgrad_kernel<*> (dense conv weight gradients)", None),
    "ss2d_scan_bwd": ("ss2d_scan_bwd", "hbm", "ss2d_scan_bwd_kernel", None),
    "dwact_bwd": ("dwact_bwd", "hbm", "dwact_bwd_kernel", None),
    "ln_bwd": ("ln_bwd", "hbm", "ln_bwd_kernel", None),
}


def profile_start(key: str):
    global _PROF
    if key not in _KEYS:
        raise ValueError(f"profile_start: unknown key {key}; choose from {sorted(_KEYS)}")
    op, bound, symbol, pred = _KEYS[key]
    _PROF = {"kernel": op, "symbol": symbol, "bound": bound, "pred": pred, "events": [], "bytes": 0.0, "flops": 0.0}


def profile_stop():
    global _PROF
    p, _PROF = _PROF, None
    if p is None:
        return None
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in p["events"])
    return {"kernel": p["symbol"], "bound": p["bound"], "launches": len(p["events"]), "ms": ms,
            "bytes": p["bytes"], "flops": p["flops"]}


def _timed(name, nbytes, nflops):
    """Decorator-free helper: returns (start_event or None); caller records the end with _timed_end."""
    if _PROF is None or _PROF["kernel"] != name:
        return None
    s = torch.cuda.Event(enable_timing=True)
    s.record()
    _PROF["bytes"] += nbytes
    _PROF["flops"] += nflops
    return s


def _timed_end(s):
    if s is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        _PROF["events"].append((s, e))


def _wrap_profiled():
    """Wrap the four hot ops with event brackets + algorithmic byte/flop accounting."""
    global pw_gemm, conv2d, dwconv3x3, ss2d_scan, transpose_planes
    _pw, _cv, _dw, _ss, _tp = pw_gemm, conv2d, dwconv3x3, ss2d_scan, transpose_planes

    def pw_gemm_p(x1, Wp, M, **kw):
        if _PROF is None:
            return _pw(x1, Wp, M, **kw)
        B, C1 = x1.shape[0], x1.shape[1]
        L = x1[0, 0].numel()
        x2, mode = kw.get("x2"), kw.get("in_mode", 0)
        K = C1 + (x2.shape[1] if mode == 2 else 0)
        cin = C1 + (x2.shape[1] if x2 is not None else 0)
        nb = 4.0 * B * L * (cin + M + (M if kw.get("res") is not None else 0)) + 4.0 * Wp.numel()
        if _PROF["kernel"] == "pw_gemm" and not _PROF["pred"](K, M, kw.get("ln") is not None, L, mode):
            return _pw(x1, Wp, M, **kw)
        s = _timed("pw_gemm", nb, 2.0 * M * K * L * B)
        out = _pw(x1, Wp, M, **kw)
        _timed_end(s)
        return out

    def conv2d_p(x, w, bias=None, stride=1, pad=1, **kw):
        if _PROF is None:
            return _cv(x, w, bias, stride, pad, **kw)
        B, _, H, W = x.shape
        Co, Ci, KH, KW = w.shape
        Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        nres = (kw.get("res1") is not None) + (kw.get("res2") is not None)
        s = _timed("conv2d", 4.0 * B * (Ci * H * W + Co * Ho * Wo * (1 + nres)) + 4.0 * w.numel(), 2.0 * B * Co * Ci * KH * KW * Ho * Wo)
        out = _cv(x, w, bias, stride, pad, **kw)
        _timed_end(s)
        return out

    def dwconv3x3_p(x, w, bias=None, mode=0):
        if _PROF is None:
            return _dw(x, w, bias, mode)
        B, Cin, H, W = x.shape
        Cout = Cin // 2 if mode == 2 else Cin
        s = _timed("dwconv3x3", 4.0 * B * H * W * (Cin + Cout), 18.0 * B * Cin * H * W)
        out = _dw(x, w, bias, mode)
        _timed_end(s)
        return out

    def ss2d_scan_p(x0, x1, xd0, xd1, dtw, dtb, A, Ds):
        if _PROF is None:
            return _ss(x0, x1, xd0, xd1, dtw, dtb, A, Ds)
        s = _timed("ss2d_scan", 4.0 * (2 * x0.numel() + 2 * xd0.numel() + 2 * x0.numel()), 0.0)
        out = _ss(x0, x1, xd0, xd1, dtw, dtb, A, Ds)
        _timed_end(s)
        return out

    def transpose_planes_p(x):
        if _PROF is None:
            return _tp(x)
        s = _timed("transpose_planes", 8.0 * x.numel(), 0.0)
        out = _tp(x)
        _timed_end(s)
        return out

    pw_gemm, conv2d, dwconv3x3, ss2d_scan, transpose_planes = pw_gemm_p, conv2d_p, dwconv3x3_p, ss2d_scan_p, transpose_planes_p

    # backward ops: (algorithmic bytes, flops) of one launch from the call's arguments
    def numel(t):
        return 0 if t is None else t.numel()

    def wrap(name, cost):
        inner = globals()[name]

        def wrapped(*a, **kw):
            if _PROF is None or _PROF["kernel"] != name:
                return inner(*a, **kw)
            nb, nf = cost(*a, **kw)
            s = _timed(name, nb, nf)
            out = inner(*a, **kw)
            _timed_end(s)
            return out
        wrapped.__doc__ = inner.__doc__
        globals()[name] = wrapped

    def c_pw_wgrad(dy, x1, dw, x2=None, dbias=None, blk_rows=0, perm=None, dy_bstride=0, M=None):
        B, L = x1.shape[0], x1[0, 0].numel()
        M = dy.shape[1] if M is None else M
        K = x1.shape[1] + (0 if x2 is None else x2.shape[1])
        return 4.0 * B * L * (M + K) + 4.0 * M * K, 2.0 * M * K * B * L

    def c_conv_wgrad(dy, x, dw, dbias=None, stride=1, pad=1, cin_slice=None):
        Cout, Cin, KH, KW = dw.shape
        return 4.0 * (dy.numel() + x.shape[0] * Cin * x.shape[2] * x.shape[3]) + 4.0 * dw.numel(), 2.0 * dy.numel() * Cin * KH * KW

    def c_scan_bwd(x0, x1, xd0, xd1, dy0, dy1, *a, **kw):
        return 4.0 * (6 * x0.numel() + 4 * xd0.numel()), 0.0

    def c_dwact(t, w, bias, dout, dw, dbias, mode):
        return 4.0 * (2 * t.numel() + dout.numel()), 40.0 * t.numel()

    def c_ln(x1, dn, gamma, beta, eps, dgamma, dbeta, x2=None, dres=None, want_n=True):
        return 4.0 * x1.numel() * (3 + (x2 is not None) + (dres is not None) + bool(want_n)), 0.0

    for nm, fn in (("pw_wgrad_", c_pw_wgrad), ("conv_wgrad_", c_conv_wgrad), ("ss2d_scan_bwd", c_scan_bwd), ("dwact_bwd", c_dwact), ("ln_bwd", c_ln)):
        wrap(nm, fn)


_wrap_profiled()
