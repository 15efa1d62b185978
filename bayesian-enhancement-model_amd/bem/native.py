"""ctypes binding of libbem_hip.so (C ABI declared in include/bem_hip.h).

There is deliberately no fallback: if the library is missing or a call is rejected the caller
gets an exception (``BemNativeError``) -- the product path never computes on the CPU."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BEM_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libbem_hip.so"))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")


class BemNativeError(RuntimeError):
    pass


class PwArgs(ctypes.Structure):
    """Mirror of ``bem_pw_args`` (include/bem_hip.h)."""
    _fields_ = [
        ("x1", c_void_p), ("x2", c_void_p), ("C1", c_int), ("C2", c_int), ("in_mode", c_int),
        ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_eps", c_float),
        ("Wp", c_void_p), ("w_bstride", c_int64),
        ("bias", c_void_p), ("bias_bstride", c_int64),
        ("res", c_void_p),
        ("prelu", c_void_p), ("act", c_int),
        ("out", c_void_p), ("out_mode", c_int), ("Win", c_int),
        ("B", c_int), ("M", c_int), ("K", c_int), ("L", c_int),
    ]


class WgradArgs(ctypes.Structure):
    """Mirror of ``bem_wgrad_args`` (include/bem_hip.h)."""
    _fields_ = [
        ("dy", c_void_p), ("dy_bstride", c_int64), ("M", c_int),
        ("x1", c_void_p), ("x1_bstride", c_int64), ("C1", c_int),
        ("x2", c_void_p), ("x2_bstride", c_int64), ("C2", c_int),
        ("dw", c_void_p), ("ldw", c_int64), ("blk_rows", c_int), ("perm", c_int * 4),
        ("dbias", c_void_p),
        ("B", c_int), ("L", c_int),
    ]


P, I, I64, U64, F = c_void_p, c_int, c_int64, c_uint64, c_float

# name -> argtypes (return type int unless listed in _RESTYPE); this table is what the
# "every declared symbol is exported" CPU test checks against include/bem_hip.h.
SIGNATURES = {
    "bem_selective_scan_fwd_f32": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "bem_selective_scan_fwd_in16": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "bem_cast16_to_f32": [P, P, ctypes.c_int64, I, P],
    "bem_cast_f32_to16": [P, P, ctypes.c_int64, I, P],
    "bem_selective_scan_bwd_ws_elems": [I, I, I, I],
    "bem_selective_scan_bwd_f32": [P] * 16 + [I, I, I, I, I, I, P],
    "bem_cross_scan_f32": [P, P, I, I, I, I, P],
    "bem_cross_merge_f32": [P, P, I, I, I, I, P],
    "bem_ss2d_scan_f32": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "bem_ss2d_scan_strided_f32": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I64, I64, P],
    "bem_ss2d_scan_rm_supported": [I, I, I],
    "bem_ss2d_scan_rm_f32": [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I64, I64, P],
    "bem_pack_pw_weight_f32": [P, P, I, I, I, P],
    "bem_pw_packed_elems": [I, I],
    "bem_pw_gemm_x6_f32": [ctypes.POINTER(PwArgs), P],
    "bem_pack_pw_weight_x6": [P, P, I, I, I, P],
    "bem_pack_pw_weight_x6_strided": [P, P, I, I, I, I64, I64, I64, P],
    "bem_pack_pw_weight_x6_jobs": [P, P, I, P, P],
    "bem_pw_x6_packed_elems": [I, I],
    "bem_bnn_sample_pack_x6": [P, P, P, P, I, I, I, U64, U64, P, I, P],
    "bem_store_words": [P, P, I, P],
    "bem_bnn_ebank_sample_f32": [P, P, I, P, U64, U64, P],
    "bem_bnn_bank_sample_f32": [P, P, I, P, P, P, P, P, F, P, U64, U64, P, P],
    "bem_bnn_bank_kl_f32": [P, P, I, P, P, P, P],
    "bem_bnn_bank_kl_bwd_f32": [P, P, I, P, P, P, P],
    "bem_bnn_bank_reparam_bwd_f32": [P, P, I, P, P, P],
    "bem_row_scale_f32": [P, P, P, I, I, P],
    "bem_se_gate_f32": [P, P, P, P, I, I, I, P],
    "bem_spatial_attention_f32": [P, P, P, P, P, I, I, I, I, I, P],
    "bem_hamilton_bwd_f32": [P, P, P, I, I, I, P],
    "bem_chan_scale_f32": [P, P, I64, P, P, F, P, I, I, I64, P],
    "bem_chan_dot_f32": [P, P, P, I, I, I64, I, P],
    "bem_se_gate_bwd_f32": [P, P, P, P, P, P, P, P, I, I, I, P],
    "bem_spatial_attention_bwd_f32": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "bem_bnn_prior_ema_f32": [P, P, P, P, F, P, I64, P],
    "bem_bnn_kl_f32": [P, P, P, P, I64, P, P],
    "bem_bnn_kl_bwd_f32": [P, P, P, P, I64, P, P, P, P],
    "bem_bnn_reparam_bwd_f32": [P, P, P, P, P, I64, P],
    "bem_mask_token_f32": [P, P, P, P, I, I, I, I, P],
    "bem_mask_token_bwd_f32": [P, P, P, P, I, I, I, I, P],
    "bem_depth_to_space_f32": [P, P, I, I, I, I, P],
    "bem_prelu_f32": [P, P, P, I64, P],
    "bem_prelu_bwd_f32": [P, P, P, P, P, I64, P],
    "bem_bilinear_up_bwd_f32": [P, P, I, I, I, I, I, P],
    "bem_gdmlp_x6_f32": [P, P, P, F, P, P, P, P, P, P, I, I, I, I, I, P],
    "bem_conv3x3_x6_f32": [P, I64, P, P, P, P, P, I, I, I, I, I, I, P],
    "bem_conv4x4s2_x6_f32": [P, I64, P, P, P, P, P, I, I, I, I, I, I, P],
    "bem_conv4x4s2_fast_supported": [I, I, I],
    "bem_conv3x3_rows_supported": [I, I, I],
    "bem_ss2d_front_x6_f32": [P, P, P, F, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "bem_conv_taps_x6_f32": [P, I64, P, P, P, P, P, I, I, I, I, I, I, I, I, I, P],
    "bem_dwconv3x3_f32": [P, P, I64, P, I64, P, I, I, I, I, I, P],
    "bem_conv2d_f32": [P, I64, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "bem_conv2d_mfma_f32": [P, I64, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "bem_quat_dwt_f32": [P, I64, P, I, I, I, P],
    "bem_dwt_f32": [P, P, I, I, I, I, P],
    "bem_iwt_f32": [P, P, I, I, I, I, P],
    "bem_iwt_hamilton_f32": [P, P, P, I, I, I, P],
    "bem_hamilton_f32": [P, P, I, I, I, P],
    "bem_hamilton_full_f32": [P, P, P, I, I, I, P],
    "bem_attn_stats_f64": [P, P, P, I, I, P],
    "bem_attn_fold_f32": [P, P, P, P, P, P, I, I, P],
    "bem_transpose_planes_f32": [P, I64, P, I64, I, I, I, I, P],
    "bem_copy_channels_f32": [P, I64, P, I64, I, I, I, P],
    "bem_copy_channels_rep_f32": [P, I64, P, I64, I, I, I, I, P],
    "bem_add_channels_f32": [P, I64, P, I64, I, I, I, P],
    "bem_bilinear_up_f32": [P, I64, P, I64, I, I, I, I, I, P],
    "bem_space_to_depth_f32": [P, P, I, I, I, I, P],
    "bem_pixel_shuffle2_f32": [P, P, I, I, I, I, P],
    "bem_bnn_sample_f32": [P, P, P, P, I, I64, U64, U64, P, P],
    "bem_select_best_f32": [P, P, P, P, P, I, I, I64, P],
    "bem_ssim_f32": [P, P, P, P, I, I, I, I, P],
    "bem_select_scores_f32": [P, P, P, F, I, P, P, P, P, I, I, I64, P],
    "bem_mc_mean_f32": [P, P, P, P, I, I, I, I, I, I, I, P],
    "bem_pad_reflect_f32": [P, P, I, I, I, I, I, P],
    "bem_resize_down_f32": [P, P, I, I, I, I, P],
    "bem_randn_f32": [P, I64, U64, U64, P, P],
    "bem_cond_postproc_f32": [P, P, P, P, I, I, I, I, F, P],
    "bem_plane_mean_f32": [P, P, I, I, I, I, I, P],
    "bem_candidate_finalize_f32": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "bem_l1_loss_f32": [P, P, P, P, P, I64, F, P, P],
    "bem_iwt_hamilton_bwd_f32": [P, P, P, P, P, I, I, I, P],
    "bem_pixel_unshuffle2_f32": [P, P, I, I, I, I, P],
    "bem_channel_sum_f32": [P, P, I, I, I64, P],
    "bem_add_f32": [P, P, P, I64, F, P],
    "bem_ln_bwd_f32": [P, P, P, P, P, F, P, P, P, P, P, I, I, I64, P],
    "bem_ln_fwd_f32": [P, P, P, P, F, P, I, I, I64, P],
    "bem_dwact_bwd_f32": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "bem_pw_wgrad_f32": [ctypes.POINTER(WgradArgs), P],
    "bem_pw_wgrad_x6_f32": [ctypes.POINTER(WgradArgs), P, I64, P],
    "bem_pw_wgrad_x6_ws_elems": [I, I, I, I],
    "bem_conv_wgrad_f32": [P, P, I64, P, P, I, I, I, I, I, I, I, I, I, P],
    "bem_ss2d_scan_bwd_f32": [P] * 18 + [I, I, I, I, I64, I64, P],
    "bem_grad_sumsq_f32": [P, I64, P, P],
    "bem_adamw_step_f32": [P, P, P, P, I64, F, F, F, F, F, I, F, P, P, P, P],
    "bem_last_error": [],
    "bem_abi_version": [],
}
_RESTYPE = {"bem_last_error": ctypes.c_char_p, "bem_pw_packed_elems": c_int64, "bem_pw_x6_packed_elems": c_int64, "bem_selective_scan_bwd_ws_elems": c_int64,
            "bem_pw_wgrad_x6_ws_elems": c_int64}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into lib/libbem_hip.so (hipcc cross-compiles without a GPU)."""
    import subprocess
    r = subprocess.run(["make", "-j8", "-C", CSRC], capture_output=True, text=True)
    if r.returncode != 0:
        raise BemNativeError(f"building libbem_hip.so failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(r.stdout)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BemNativeError(
                f"{LIB_PATH} not found: the HIP library is required (no CPU fallback). "
                f"Build it with `make -C {CSRC}` or __graft_entry__.build().")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError here = symbol missing from the build
            fn.argtypes = argtypes
            fn.restype = _RESTYPE.get(name, c_int)
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().bem_last_error()
        raise BemNativeError(f"{what or 'bem call'} failed (rc={rc}): {msg.decode() if msg else ''}")
