/*
 * bem_hip.h -- C ABI of libbem_hip.so: the MI355X (gfx950) hot path of the Bayesian Enhancement
 * Model (two-stage N-sample Bayesian low-light enhancement).
 *
 * Conventions (all entry points)
 *   - plain device pointers + sizes, float32, NCHW / (B, C, L) contiguous unless a stride is given;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *     no internal synchronisation, no global state, re-entrant across streams -- the same contract
 *     as the reference's extension (selective_scan_oflex.cpp:232-233);
 *   - return value 0 = launched, non-zero = rejected (bad shape / null pointer / HIP launch error);
 *     bem_last_error() gives the message for the calling thread.  Nothing is launched on rejection.
 *   - "reference" paths below are relative to vfrantc/Bayesian-Enhancement-Model.
 *
 * The library has NO CPU path: without a gfx950 device the calls fail (hipErrorNoDevice).
 */
#ifndef BEM_HIP_H
#define BEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEM_OK 0
#define BEM_ERR_INVALID 1
#define BEM_ERR_LAUNCH 2

const char* bem_last_error(void);
int bem_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Operator seam: what basicsr/vmamba/models/csms6s.py and csm_triton.py call.
 * ------------------------------------------------------------------------------------------- */

/* Replaces selective_scan_cuda_oflex.fwd (kernels/selective_scan/csrc/selective_scan/cusoflex/
 * selective_scan_oflex.cpp:157-243; kernel selective_scan_fwd_kernel_oflex.cuh:67-180), f32 in/out.
 * u, delta, out: (batch, dim, L); A: (dim, dstate); Bm, Cm: (batch, ngroups, dstate, L);
 * D, delta_bias: (dim) or NULL.  dim % ngroups == 0, 1 <= dstate <= 256. */
int bem_selective_scan_fwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                               const float* Cm, const float* D, const float* delta_bias, float* out,
                               int batch, int dim, int L, int dstate, int ngroups, int delta_softplus,
                               void* stream);

/* The same call with u, delta, Bm, Cm in float16 (in_dtype 1) or bfloat16 (in_dtype 2), read as they are (8-byte aligned when
 * L % 4 == 0); A, D, delta_bias and out float32 -- the extension's input_t = at::Half / at::BFloat16 instantiations with
 * out_float = true (selective_scan_oflex.cpp:166-216; csms6s.py:85 passes oflex). */
int bem_selective_scan_fwd_in16(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm, const float* D,
                                const float* delta_bias, float* out, int in_dtype, int batch, int dim, int L, int dstate,
                                int ngroups, int delta_softplus, void* stream);

/* Element casts for the 16-bit forms of the seam's backward (the extension converts inside its kernels, selective_scan_bwd_kernel_oflex.cuh:
 * 108-140; here the f32 kernel runs between two casts): dtype 1 = float16, 2 = bfloat16, round to nearest even, n elements. */
int bem_cast16_to_f32(const void* src, float* dst, int64_t n, int dtype, void* stream);
int bem_cast_f32_to16(const float* src, void* dst, int64_t n, int dtype, void* stream);

/* Replaces selective_scan_cuda_oflex.bwd (selective_scan_oflex.cpp:245-358, kernel selective_scan_bwd_kernel_oflex.cuh:73-289),
 * f32.  dout, du, ddelta: (batch, dim, L); dA (dim, dstate); dB, dC (batch, ngroups, dstate, L) f32; dD, ddelta_bias (dim)
 * or NULL exactly when D / delta_bias are NULL.  ws: scratch of bem_selective_scan_bwd_ws_elems(...) floats (the role of
 * the reference's opaque `x` blob; recomputed here, so the forward need not be re-run by the caller).  dA/dB/dC/dD/
 * ddelta_bias are zeroed by the call and accumulated with float atomics across batch / group members (run-to-run
 * last-bit differences, like the reference). */
int64_t bem_selective_scan_bwd_ws_elems(int batch, int dim, int L, int dstate);
int bem_selective_scan_bwd_f32(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                               const float* D, const float* delta_bias, const float* dout, float* ws, float* du,
                               float* ddelta, float* dA, float* dB, float* dC, float* dD, float* ddelta_bias,
                               int batch, int dim, int L, int dstate, int ngroups, int delta_softplus, void* stream);

/* Replaces cross_scan_fn / triton_cross_scan_flex (csm_triton.py:278-423): x (B,C,H,W) -> xs (B,4,C,H*W). */
int bem_cross_scan_f32(const float* x, float* xs, int B, int C, int H, int W, void* stream);
/* Replaces cross_merge_fn (csm_triton.py:446-471): ys (B,4,C,H,W) -> y (B,C,H*W). */
int bem_cross_merge_f32(const float* ys, float* y, int B, int C, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused SS2D core (replaces the cross_scan -> x_proj -> dt_proj -> selective_scan chain of
 * SS2D.forward_corev2, basicsr/vmamba/models/vmamba.py:657-684) for d_state = 1, K = 4.
 *
 *   x0  (B,C,L)        activations in row-major pixel order (directions 0 fwd / 2 rev)
 *   x1  (B,C,L)        the same planes transposed (W,H) (directions 1 fwd / 3 rev)
 *   xd0 (B,2,R+2,L)    x_proj output rows [dt_0..dt_{R-1}, B, C] of directions {0,2}, row-major order
 *   xd1 (B,2,R+2,L)    the same for directions {1,3}, transposed order
 *   dtw (4,C,R), dtb (4,C), A (4C) = -exp(A_logs), Ds (4C)
 *   y0  (B,C,L)        out: y(dir0) + y(dir2) in row-major order
 *   y1  (B,C,L)        out: y(dir1) + y(dir3) in transposed order
 * ------------------------------------------------------------------------------------------- */
int bem_ss2d_scan_f32(const float* x0, const float* x1, const float* xd0, const float* xd1,
                      const float* dtw, const float* dtb, const float* A, const float* Ds,
                      float* y0, float* y1, int B, int C, int L, int R, void* stream);
/* The same with x_dbl tensors that are channel slices of wider buffers: xd*_bstride = elements between batch rows
 * (0 = contiguous 2*(R+2)*L; multiples of 4).  Lets one x_proj GEMM over the row-major planes produce all four
 * directions' rows, the {1,3} half being transposed afterwards (10 planes instead of a second pass over C planes). */
int bem_ss2d_scan_strided_f32(const float* x0, const float* x1, const float* xd0, const float* xd1,
                              const float* dtw, const float* dtb, const float* A, const float* Ds,
                              float* y0, float* y1, int B, int C, int L, int R, int64_t xd0_bstride, int64_t xd1_bstride,
                              void* stream);

/* Row-major in, row-major out: x (B,C,H,W) feeds both orientations and y1 (directions 1 + 3) comes back already in
 * (B,C,H,W) order -- a workgroup of the column-major orientation stages its planes in LDS, so neither the transposed copy
 * of x nor the transposed y exist in HBM.  xd1 stays in transposed pixel order (2 (R+2) small planes).  Plane sizes / ranks
 * of a 256x256 image only: bem_ss2d_scan_rm_supported(H, W, R) != 0. */
int bem_ss2d_scan_rm_supported(int H, int W, int R);
int bem_ss2d_scan_rm_f32(const float* x, const float* xd0, const float* xd1, const float* dtw, const float* dtb,
                         const float* A, const float* Ds, float* y0, float* y1, int B, int C, int H, int W, int R,
                         int64_t xd0_bstride, int64_t xd1_bstride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise (1x1) channel-mix GEMM with fused prologue / epilogue (argument block of bem_pw_gemm_x6_f32).  Replaces
 * Linear2d / nn.Conv2d(k=1) / LayerNorm2d+Linear2d chains (vmamba.py:42-63,123-133,702,715,1326-1334).
 *
 *   out[b][m][p] = act( sum_k W[b?][m][k] * pro(x)[b][k][p] + bias[b?][m] ) + res[b][m][p]
 *   pro: in_mode 0: x = x1 (K = C1) | 1: x = x1 + x2 (K = C1 = C2) | 2: x = cat(x1, x2) (K = C1 + C2);
 *        then LayerNorm over the K channels of each pixel when ln_w != NULL (eps = ln_eps).
 *   Wp : weights pre-packed by bem_pack_pw_weight_x6; w_bstride / bias_bstride = element stride
 *        between per-batch-element weight sets (0 = shared by the whole batch).
 *   act: 0 none, 1 PReLU with the single slope *prelu.
 *   out_mode 0: out (B,M,L).  out_mode 1: ConvTranspose2d(k=2,s=2) scatter -- M = 4*Co, row
 *        (dy*2+dx)*Co + co goes to out (B,Co,2H,2W)[co][2y+dy][2x+dx], L = H*Win.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    const float* x1; const float* x2; int C1; int C2; int in_mode;
    const float* ln_w; const float* ln_b; float ln_eps;
    const float* Wp; int64_t w_bstride;
    const float* bias; int64_t bias_bstride;
    const float* res;
    const float* prelu; int act;
    float* out; int out_mode; int Win;
    int B; int M; int K; int L;
} bem_pw_args;
/* natural (nsets, M, K) row-major -> packed f32-MFMA operand order (nsets, MT, KS, 64), MT = ceil(M/32), KS = ceil(K/2):
 * the weight format of the implicit-GEMM convolutions (bem_conv2d_mfma_f32). */
int bem_pack_pw_weight_f32(const float* W, float* Wp, int nsets, int M, int K, void* stream);
int64_t bem_pw_packed_elems(int M, int K);

/* The same GEMM contract (bem_pw_args, identical semantics and error behaviour) with the products evaluated on the
 * bf16 matrix cores as an exact 3-limb expansion of both f32 operands (six limb products per term, f32 accumulate;
 * the dropped terms are <= 2^-23 of a product, i.e. f32-level error) -- 6/16 of the f32-MFMA cost.
 * Wp must come from bem_pack_pw_weight_x6: natural (nsets, M, K) f32 -> (nsets, MT, KB, 3 limbs, 64 lanes) 16-byte vectors
 * of 8 bf16, MT = ceil(M/32), KB = ceil(K/16); bem_pw_x6_packed_elems(M, K) = floats per set (w_bstride unit), 16-byte
 * aligned.  LayerNorm prologue for K <= 160. */
int bem_pw_gemm_x6_f32(const bem_pw_args* a, void* stream);
int bem_pack_pw_weight_x6(const float* W, float* Wp, int nsets, int M, int K, void* stream);
/* The same from a strided view (element strides over set, row, k): the transposed weight of an input-gradient GEMM (W^T = strides (., 1, K))
 * or a column block of a concatenated weight is packed where it lies, without a contiguous copy. */
int bem_pack_pw_weight_x6_strided(const float* W, float* Wp, int nsets, int M, int K, int64_t set_stride, int64_t row_stride,
                                  int64_t col_stride, void* stream);
/* Many matrices packed by one launch.  jobs: njobs x 8 64-bit words {source pointer; first float of the packed output in `arena` (multiple
 * of 4); int32 M, int32 K; row stride; column stride (elements); work items = ceil(M/32) * ceil(K/16) * 64; 0; 0}; blks: nblk x {int32 job,
 * int32 first block of 256 work items}.  Each output is what bem_pack_pw_weight_x6_strided writes for that view (one set). */
int bem_pack_pw_weight_x6_jobs(const void* jobs, const void* blks, int nblk, float* arena, void* stream);
int64_t bem_pw_x6_packed_elems(int M, int K);
/* bem_bnn_sample_f32 (below) and bem_pack_pw_weight_x6 in one pass for the weights of a Bayesian 1x1 layer: nsets weight
 * sets w = mu + log1p(exp(rho)) * eps written straight in x6 operand order; eps (nsets, M, K) injected or NULL = the
 * sampler's Philox draws for (seed, stream_id) -- identical values to sampling first and packing afterwards. */
int bem_bnn_sample_pack_x6(const float* mu, const float* rho, const float* eps, float* Wp, int nsets, int M, int K,
                           uint64_t seed, uint64_t stream_id, const uint64_t* stream_add, int sigma_given,
                           void* stream);   /* sigma_given: rho already holds log1p(exp(rho)); stream_add: see bem_bnn_sample_f32 */

/* ---------------------------------------------------------------------------------------------
 * Convolutions.
 * ------------------------------------------------------------------------------------------- */
/* Depthwise 3x3, padding 1 (vmamba.py:507-515 conv2d, :124 gdMlp.dwconv, QD/model4.py:157-165).
 *   mode 0: out[c] = dw(x[c]) (+bias)            mode 1: SiLU(...)
 *   mode 2: gdMlp gate, x has 2*Cout channels: out[c] = GELU(dw(x[c])) * dw(x[c+Cout])
 *   mode 3: PostSmooth: out[c] = x[c] + ReLU(dw(x[c]) + bias)
 * w: (Cw,1,3,3), bias (Cw) or NULL, Cw = 2*Cout in mode 2 else Cout; *_bstride as above. */
int bem_dwconv3x3_f32(const float* x, const float* w, int64_t w_bstride, const float* bias,
                      int64_t bias_bstride, float* out, int B, int Cout, int H, int W, int mode, void* stream);

/* Dense direct convolution KHxKW, given stride / zero padding (nn.Conv2d semantics), optional bias,
 * ReLU and up to two residual tensors added after the activation:
 *   out = relu?(conv(x) + bias) + res1 + res2.      x (B,Cin,H,W) -> out (B,Cout,Ho,Wo)
 * x may be a channel slice of a wider tensor: x_bstride = elements between batch items. */
int bem_conv2d_f32(const float* x, int64_t x_bstride, const float* w, const float* bias, const float* res1,
                   const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                   int stride, int pad, int relu, void* stream);

/* The same convolution as an implicit GEMM on the f32 matrix cores; Wp = the (Cout, Cin*KH*KW) view of the weight
 * packed by bem_pack_pw_weight_f32.  Cout <= 160; 3x3 stride 1 and 4x4 stride 2. */
int bem_conv2d_mfma_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1,
                        const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int KH, int KW,
                        int stride, int pad, int relu, void* stream);

/* The 3x3 stride-1 pad-1 case as nine shifted 1x1 taps on the bf16-limb GEMM (see bem_pw_gemm_x6_f32): no im2col patch,
 * f32-level error.  Wp = bem_pack_pw_weight_x6 of the (9, Cout, Cin) tap matrices, tap = ky*3 + kx (nsets = 9).
 * W even, Cin % 8 == 0; x_bstride as above; out = relu?(conv + bias) + res1 + res2. */
int bem_conv3x3_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1,
                       const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int relu, void* stream);
/* The 4x4 stride-2 pad-1 down-sampling convolution (DecompDualBranchDDWavelet_arch.py:40-41) on the same machinery, tap = ky*4 + kx; even output
 * width, Cin % 8 == 0.  Shapes bem_conv4x4s2_fast_supported accepts (W = 2 Wo with Wo a power of two <= 64, even H; x and x_bstride 16-byte
 * aligned; no residual inputs) run the coalesced-row kernel of conv4_x6.hip: one aligned 16-byte load per lane, channel and input row, the two
 * outer columns from the neighbour lanes, tap weights staged in LDS by LDS-DMA; other shapes run the 16 shifted taps. */
int bem_conv4x4s2_fast_supported(int Cin, int H, int W);
/* 1 when bem_conv3x3_x6_f32 runs its row form (conv_rows_x6.hip: W a power of two in 4 .. 128, Cin % 8 == 0, 16-byte aligned tensors), the
 * same kernel family with four output pixels per lane; other shapes run the nine shifted taps. */
int bem_conv3x3_rows_supported(int Cin, int H, int W);
int bem_conv4x4s2_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1,
                         const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int relu, void* stream);

/* The general entry of the tap form: K x K taps with `stride` and `dilation`; supported: 3x3 s1 d1 (pad 1), 3x3 s1 d2 (pad 2: the dilated
 * branch convolutions of QD/model2.py:171-181), 3x3 s2 d1 (pad 1: down_conv of QD/model3.py:176), 4x4 s2 d1 (pad 1).  Wp as above
 * (K*K tap matrices, tap = ky*K + kx). */
int bem_conv_taps_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1, const float* res2,
                         float* out, int B, int Cin, int H, int W, int Cout, int K, int stride, int dilation, int relu, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Quaternion / Haar primitives (basicsr/QD/model4.py:7-37,216-232; QD/quaternion.py:3-17).
 * ------------------------------------------------------------------------------------------- */
/* RGB (B,3,H,W; x_bstride elements between batch items) -> quaternion stack (8 ch) -> Haar DWT
 * -> out (B,32,H/2,W/2), channel = band*8 + q, bands LL,HL,LH,HH. */
int bem_quat_dwt_f32(const float* rgb, int64_t x_bstride, float* out, int B, int H, int W, void* stream);
int bem_dwt_f32(const float* x, float* out, int B, int C, int H, int W, void* stream);
int bem_iwt_f32(const float* x, float* out, int B, int C4, int H, int W, void* stream);
/* IWT of q1w, q2w (B,16,h,w) + Hamilton product, real part dropped -> out (B,3,2h,2w).
 * (DecompDualBranchDDWavelet_arch.py:361-367) */
int bem_iwt_hamilton_f32(const float* q1w, const float* q2w, float* out, int B, int h, int w, void* stream);
/* Hamilton product of q[:, 0:4] and q[:, 4:8] (B,8,H,W), real part dropped -> (B,3,H,W). */
int bem_hamilton_f32(const float* q, float* out, int B, int H, int W, void* stream);

/* hamilton_product(q1, q2) of QD/quaternion.py:3-17: q1, q2 (B,4,H,W) -> out (B,4,H,W) = [real, i, j, k]. */
int bem_hamilton_full_f32(const float* q1, const float* q2, float* out, int B, int H, int W, void* stream);

/* Channel cross-attention of the decomposition net (QD/model4.py:81-139) folded with the 1x1 `fuse`
 * conv that follows it.  Step 1: accumulate per image S = F1 F2^T (32x32), s1 = F1 1, s2 = F2 1 in f64
 * (stats: (B, 32*32 + 64) doubles, zeroed by the call).  Step 2: softmax + fold all 1x1 weights into
 * one per-image (32 x 64) matrix + bias: W_out (B, 32, 64) row-major, columns 0..31 for f1, 32..63 for f2 (bem_pw_gemm_x6_f32 in_mode 2
 * after bem_pack_pw_weight_x6). */
int bem_attn_stats_f64(const float* f1, const float* f2, double* stats, int B, int L, void* stream);
int bem_attn_fold_f32(const double* stats, const float* attn_w /* 8 x (32x32 + 32): q1,k2,v2,q2,k1,v1,out1,out2 */,
                      const float* fuse_w /* (32,64) */, const float* fuse_b /* (32) */,
                      float* W_out /* (B, 32, 64) */, float* bias_out /* (B,32) */, int B, int L,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Layout / resampling helpers.
 * ------------------------------------------------------------------------------------------- */
/* (P planes of H x W) -> (P planes of W x H); src/dst plane p at base + (p / ppb) * bstride + (p % ppb) * H*W. */
int bem_transpose_planes_f32(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride,
                             int nbatch, int ppb, int H, int W, void* stream);
/* dst[b][dst_c0 + c][l] = src[b][c][l]  (c < C), strides in elements. */
int bem_copy_channels_f32(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride,
                          int B, int C, int L, void* stream);
/* The same with dst row b reading src row b / rep (B = dst rows): decomp(image), evaluated once per image, handed to its rep
 * Monte-Carlo samples (the loop of eval.py:199-213 feeds the same image to every sample). */
int bem_copy_channels_rep_f32(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride, int B, int C, int L, int rep,
                              void* stream);
/* dst[b][c][l] += src[b][c][l]  (c < C); DecompDualBranch2's "Q + [cond, 0]" (DecompDualBranch_arch.py:241-246). */
int bem_add_channels_f32(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride,
                         int B, int C, int L, void* stream);
/* F.interpolate(scale_factor=s, mode='bilinear', align_corners=False) (eval.py:220, UNet_arch.py:130). */
int bem_bilinear_up_f32(const float* src, int64_t src_bstride, float* dst, int64_t dst_bstride,
                        int B, int C, int H, int W, int s, void* stream);
/* PatchMerging gather (UNet_arch.py:74-78): (B,C,H,W) -> (B,4C,H/2,W/2), blocks [ee, oe, eo, oo]. */
int bem_space_to_depth_f32(const float* x, float* out, int B, int C, int H, int W, void* stream);
/* nn.PixelShuffle(2): (B,4C,H,W) -> (B,C,2H,2W). */
int bem_pixel_shuffle2_f32(const float* x, float* out, int B, int C, int H, int W, void* stream);

/* Every Bayesian tensor of a net drawn for a stochastic forward in one launch (the N weight sets per leaf of eval.py:199-211, all leaves).
 * segs: nseg x 8 64-bit words {mu, sig pointers (sig = precomputed sigma for packed tensors, rho for natural ones); first float of the
 * tensor's output in `arena` (multiple of 4); n = elements per set; int32 M, int32 K (K > 0: the nsets sets in x6 operand order exactly as
 * bem_bnn_sample_pack_x6 with sigma_given writes them; K = 0: natural order as bem_bnn_sample_f32); stream counter; work items (packed:
 * nsets * ceil(M/32) * ceil(K/16) * 64, natural: ceil(nsets * n / 4)); nsets * n}; blks: nblk x {int32 segment, int32 first block of 256
 * work items}.  Draws: (seed, stream_base + counter), element numbering of the per-tensor entry points. */
int bem_bnn_ebank_sample_f32(const void* segs, const void* blks, int nblk, float* arena, uint64_t seed, uint64_t stream_base, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DecompDualBranch's bottleneck blocks (basicsr/archs/DecompModel_arch.py:57-99).
 * ------------------------------------------------------------------------------------------- */
/* out[m][k] = w[m][k] * scale[m]: folds CrossFusionBlock's per-channel gate (:57-66, x_tgt + gate * (W x_src + b)) into W (M,K) and,
 * with K = 1, into b; the block then runs as bem_pw_gemm_x6_f32 with x_tgt as the residual. */
int bem_row_scale_f32(const float* w, const float* scale, float* out, int M, int K, void* stream);
/* SEBlock's gate (:68-83): y (B,C) = sigmoid(w2 relu(w1 mean)), mean (B,C) from bem_plane_mean_f32, w1 (Cr,C), w2 (C,Cr), no biases. */
int bem_se_gate_f32(const float* mean, const float* w1, const float* w2, float* y, int B, int C, int Cr, void* stream);
/* SpatialAttention (:85-99) of x * chan_scale (chan_scale (B,C) = the SE gate that precedes it, :318-324, or NULL):
 * out = x * chan_scale * sigmoid(conv_kxk([mean_c, max_c](x * chan_scale))), w (1,2,k,k), k = 3 or 7, zero padding k/2, no bias;
 * map_ws: (B,2,H,W) floats of workspace; x, out (B,C,H,W). */
int bem_spatial_attention_f32(const float* x, const float* chan_scale, const float* w, float* map_ws, float* out, int B, int C, int H, int W,
                              int k, void* stream);

/* Training side of the same blocks (autograd nodes of bem/autograd.py: CrossFusion = PwFn + GateAddFn, SEBlockFn, SpatialAttnFn).
 * chan_scale: out = scale[b * scale_bstride + c] * x (+ add) (+ add_bc[b][c] * add_bc_scale); scale_bstride = C (per-image factors) or 0
 * (a parameter).  chan_dot: per_batch ? out (B,C) = sum_p a b : out (C) += sum_{b,p} a b.  se_gate_bwd: backward of bem_se_gate_f32 through
 * sigmoid / W2 / relu / W1 (dw1, dw2 accumulated, dmean (B,C) written).  spatial_attention_bwd: backward of bem_spatial_attention_f32 with
 * chan_scale = NULL; map = the (B,2,H,W) workspace its forward filled, dpre_ws (B,H,W) floats, dw (1,2,k,k) accumulated. */
int bem_chan_scale_f32(const float* x, const float* scale, int64_t scale_bstride, const float* add, const float* add_bc, float add_bc_scale,
                       float* out, int B, int C, int64_t HW, void* stream);
int bem_chan_dot_f32(const float* a, const float* b, float* out, int B, int C, int64_t HW, int per_batch, void* stream);
int bem_se_gate_bwd_f32(const float* mean, const float* w1, const float* w2, const float* y, const float* dy, float* dmean, float* dw1,
                        float* dw2, int B, int C, int Cr, void* stream);
int bem_spatial_attention_bwd_f32(const float* x, const float* dout, const float* map, const float* w, float* dpre_ws, float* dx, float* dw,
                                  int B, int C, int H, int W, int k, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bayesian sampling + Monte-Carlo loop pieces (basicsr/bayesian/conv.py:106-114, eval.py:199-264).
 * ------------------------------------------------------------------------------------------- */
/* out[s][i] = mu[i] + log1p(exp(rho[i])) * eps, eps = eps_in[s][i] when eps_in != NULL else a
 * Philox4x32-10 N(0,1) draw keyed by (seed, stream_id, s*n + i).  stream_add (NULL or one uint64 in device memory) is added to
 * stream_id by the kernel: the per-iteration part of the id stays in HBM, so a captured HIP graph of a training step (the
 * replacement of torch's generator advancing between iterations, condition_generator_model.py:176-218) draws new numbers on
 * each replay. */
int bem_bnn_sample_f32(const float* mu, const float* rho, const float* eps_in, float* out,
                       int nsets, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* stream_add, void* stream);
/* Image preparation of eval.py:146-176: reflect-pad bottom/right of P planes (H,W) -> (Hp,Wp) (numpy 'reflect'), and the
 * x1/s INTER_LINEAR condition image of the padded planes (even s dividing Hp, Wp): mean of the 2x2 centre taps. */
int bem_pad_reflect_f32(const float* x, float* out, int P, int H, int W, int Hp, int Wp, void* stream);
int bem_resize_down_f32(const float* x, float* out, int P, int Hp, int Wp, int s, void* stream);
/* N(0,1) draws from the same Philox4x32-10 stream family as bem_bnn_sample_f32 (torch.randn_like of eval.py:209). */
int bem_randn_f32(float* out, int64_t n, uint64_t seed, uint64_t stream_id, const uint64_t* stream_add, void* stream);
/* Stage-I post-processing (eval.py:200-209): c = clamp(pred,0,1); if target_mean: c = clamp(c *
 * target_mean[b_img][ch] / mean_hw(c), 0, 1); c += noise * noise_level.  pred/out (Bn,3,h,w);
 * target_mean (n_img,3) with image index = b / samples_per_image; noise may be NULL. */
int bem_cond_postproc_f32(const float* pred, const float* target_mean, const float* noise, float* out,
                          int Bn, int h, int w, int samples_per_image, float noise_level, void* stream);
/* Per-plane mean over the top-left (h,w) window of (P, Hs, Ws) planes -> means (P) (f32 out, f64 accumulate). */
int bem_plane_mean_f32(const float* x, float* means, int P, int Hs, int Ws, int h, int w, void* stream);
/* Candidate finalisation (eval.py:222-226,246-252, Enhancement/utils.py:5-9): crop to (h,w), clamp to
 * [0,1], optional per-channel GT-mean rescale + clip, write final (Bn,3,h,w) and PSNR (Bn) vs target
 * (n_img,3,h,w).  pred is (Bn,3,Hp,Wp). */
int bem_candidate_finalize_f32(const float* pred, const float* target, float* final_out, float* psnr, double* ws,
                               int Bn, int samples_per_image, int Hp, int Wp, int h, int w, int gt_mean,
                               void* stream);   /* ws: scratch of 7*Bn doubles (zeroed by the call) */

/* Candidate selection of eval.py:284-285 (psnr_weight = 1): per image, score_i = psnr_i / max_j psnr_j, best = FIRST index
 * of the maximum score (python list.index(max(...))); best (B) int32, best_psnr (B), and -- when cand/best_img are given --
 * best_img[b] = cand[b*N + best[b]] (chw floats each).  Everything stays on the device: no host round trip per step. */
int bem_select_best_f32(const float* cand, const float* psnr, int* best, float* best_psnr, float* best_img, int B, int N,
                        int64_t chw, void* stream);

/* calculate_ssim(img_as_ubyte(target), img_as_ubyte(pred)) of Enhancement/utils.py:12-57 per candidate: uint8 values rint(255 x), f64,
 * 11x11 Gaussian (sigma 1.5) over the valid region, mean over region and channels.  pred (Bn,3,h,w), target (Bn/spi,3,h,w), h, w > 10;
 * ssim (Bn) f32 out; ws: scratch of Bn doubles (zeroed by the call). */
int bem_ssim_f32(const float* pred, const float* target, float* ssim, double* ws, int Bn, int samples_per_image, int h, int w, void* stream);

/* Selection rules of eval.py:268-297 on the device, FIRST index on ties (python list.index):
 *   rule 0: max of weight * s1 / max(s1) + (1 - weight) * s2 / max(s2)  (full reference, PSNR / SSIM; s2 NULL = weight 1)
 *   rule 1: max of s1 (no-reference, CLIP-IQA)      rule 2: min of s1 (NIQE)
 * best (B) int32; best_s1 / best_s2 (B) or NULL; best_img[b] = cand[b*N + best[b]] when cand / best_img are given. */
int bem_select_scores_f32(const float* cand, const float* s1, const float* s2, float weight, int rule, int* best, float* best_s1,
                          float* best_s2, float* best_img, int B, int N, int64_t chw, void* stream);

/* Monte-Carlo mean of eval.py:224-225,308-314: out (B,3,h,w) = clamp(mean_n clamp(pred[b*N+n][:, :h, :w], 0, 1), 0, 1), with gt_mean scaled
 * by mean(gray(target)) / mean(gray(out)) (cv2 BGR2GRAY weights on the stored channel order) and clipped.  pred (B*N,3,Hp,Wp);
 * ws: 2 B doubles (zeroed by the call) when gt_mean. */
int bem_mc_mean_f32(const float* pred, const float* target, float* out, double* ws, int B, int N, int Hp, int Wp, int h, int w,
                    int gt_mean, void* stream);

/* The whole gdMlp branch of a VSSBlock in one kernel (vmamba.py:116-133 gdMlp.forward + the block's norm2 / residual :1330-1333):
 *   out (B,C,H,W) = x + W_o (GELU(h[0:Hd]) * h[Hd:2Hd]) + b_o,   h = dw3x3(W_i LayerNorm2d(x) + b_i) + b_dw.
 * Neither the 2Hd-channel project_in output nor the Hd-channel gate tensor reaches HBM (4 x 32 pixel tiles, both live as 16-gate-channel
 * slices in LDS).  x (B,C,H,W) with C <= 80, out != x; ln_w / ln_b (C).
 * Wp_gate = bem_pack_pw_weight_x6 of the (2Hd, C) project_in matrix in gate-interleaved row order: packed row 32 j + 2 c + s =
 * W_i[s Hd + 16 j + c] (c < 16, s < 2; Hd % 16 == 0); bias_gate (Hd/16, 16, 2) = b_i in the same order (zeros for a layer without bias);
 * dw_gate10 (Hd,10,2): [c][tap < 9] = (dww[c][tap], dww[Hd + c][tap]), [c][9] = (dwb[c], dwb[Hd + c]) (zeros for a layer without bias);
 * Wp_out = bem_pack_pw_weight_x6 of the (C, Hd) project_out matrix; bias_out (C) | NULL.  Packed weights, bias_gate and dw_gate10 16-byte
 * aligned: the kernel moves them chunk by chunk into LDS with LDS-DMA (global_load_lds_dwordx4). */
int bem_gdmlp_x6_f32(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const float* Wp_gate,
                     const float* bias_gate, const float* dw_gate10, const float* Wp_out,
                     const float* bias_out, float* out, int B, int C, int Hd, int H, int W, void* stream);

/* SS2D front half in one kernel (vmamba.py:700-716 up to the scan, with the block's norm :1326):
 *   xc (B,C,H,W) = SiLU(dw3x3(W_in LayerNorm2d(x) + b_in) + b_dw),   xd (B,Mx,H*W) = W_x xc  (Mx = 4 (R + 2): the x_dbl rows of all four directions).
 * The in_proj output exists only as a 4 x 32 pixel tile (+ halo) in LDS.  x (B,C,H,W) with C <= 48, C % 8 == 0, xc != x; ln_w / ln_b (C);
 * Wp_in = bem_pack_pw_weight_x6 of the (C, C) in_proj matrix, bias_in (C) | NULL; dww (C,9), dwb (C) | NULL: depthwise 3x3;
 * Wp_x = bem_pack_pw_weight_x6 of the (Mx, C) x_proj rows, Mx <= 32. */
int bem_ss2d_front_x6_f32(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const float* Wp_in, const float* bias_in,
                          const float* dww, const float* dwb, const float* Wp_x, float* xc, float* xd, int B, int C, int Mx, int H, int W,
                          void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training step: backward of the Stage-II forward + optimizer, i.e. what
 * `l_total.backward()`, clip_grad_norm_ and AdamW.step do in basicsr/models/image_enhancer_model.py:165-216.
 * Gradient buffers of parameters (dw, dbias, dgamma, ...) are ACCUMULATED INTO (+=, float atomics) unless noted:
 * the caller zeroes them once per step (optimizer.zero_grad()).
 * ------------------------------------------------------------------------------------------- */

/* L1Loss(loss_weight, reduction='mean') (basicsr/losses/losses.py:28): loss[0] (or NULL) = weight * mean |pred - gt|;
 * dpred (or NULL) = gmul[0] * weight * sign(pred - gt) / n, gmul = device scalar dL/dloss (NULL = 1).
 * ws: one double of scratch (zeroed by the call). */
int bem_l1_loss_f32(const float* pred, const float* gt, float* dpred, float* loss, double* ws, int64_t n, float weight,
                    const float* gmul, void* stream);

/* Backward of bem_iwt_hamilton_f32: dout (B,3,2h,2w) -> dq1w, dq2w (B,16,h,w) (written). */
int bem_iwt_hamilton_bwd_f32(const float* q1w, const float* q2w, const float* dout, float* dq1w, float* dq2w, int B, int h, int w,
                             void* stream);
/* Backward of bem_hamilton_f32 (the full-resolution archs: hamilton_product(out_1, out_2)[:, 1:], DecompModel_arch.py:351-352):
 * q8 (B,8,H,W) = [p | q], dout (B,3,H,W) -> dq8 (B,8,H,W) = [dp | dq]. */
int bem_hamilton_bwd_f32(const float* q8, const float* dout, float* dq8, int B, int H, int W, void* stream);

/* nn.PixelUnshuffle(2): x (B,C,2H,2W) -> out (B,4C,H,W), out[c*4 + i*2 + j][y][x] = x[c][2y+i][2x+j] (H, W = OUTPUT plane size).
 * Rearranges dL/dout of ConvTranspose2d(k=2,s=2) so that its input / weight gradients are 1x1 GEMMs, and is the inverse of
 * bem_pixel_shuffle2_f32 (used for the input gradient of the 4x4 stride-2 convolution). */
int bem_pixel_unshuffle2_f32(const float* x, float* out, int B, int C, int H, int W, void* stream);

/* out[c] += sum over batch and pixels of x (B,C,L)  (bias gradients). */
int bem_channel_sum_f32(const float* x, float* out, int B, int C, int64_t L, void* stream);
/* out = a + alpha * b (n elements): gradient accumulation where a tensor feeds two consumers (encoder skips, shared
 * bottleneck), and conds = gt_down + noise_level * randn of the training step (image_enhancer_model.py:143-148). */
int bem_add_f32(const float* a, const float* b, float* out, int64_t n, float alpha, void* stream);

/* LayerNorm2d backward (vmamba.py:58-63): x = x1 (+ x2 when non-NULL), all (B,C,L); dn = dL/dLN(x).
 *   dx = (dres or 0) + LN'(dn)   written;   n_out (or NULL) = LN(x) written (the operand of the consumer's weight gradient);
 *   dgamma, dbeta (C) accumulated. */
int bem_ln_bwd_f32(const float* x1, const float* x2, const float* dn, const float* gamma, const float* beta, float eps,
                   const float* dres, float* dx, float* n_out, float* dgamma, float* dbeta, int B, int C, int64_t L, void* stream);
int bem_ln_fwd_f32(const float* x1, const float* x2, const float* gamma, const float* beta, float eps, float* n_out, int B, int C,
                   int64_t L, void* stream);

/* Backward of bem_dwconv3x3_f32 through the activation (modes 0 plain, 1 SiLU, 2 gdMlp gate), shared weights:
 *   t (B,Cw,H,W) the conv input, dout (B,Cout,H,W); dpre (B,Cw,H,W) = dL/d(conv output + bias) written;
 *   dw (Cw,9), dbias (Cw)|NULL accumulated.  The input gradient is bem_dwconv3x3_f32(dpre, flipped w, mode 0). */
int bem_dwact_bwd_f32(const float* t, const float* w, const float* bias, const float* dout, float* dpre, float* dw, float* dbias,
                      int B, int Cout, int H, int W, int mode, void* stream);

/* Weight gradient of a 1x1 layer: dw[m][k] += sum_{b,p} dy[b][m][p] * x[b][k][p], x = x1 rows [0,C1) then x2 rows [C1,C1+C2)
 * (the concat input mode of bem_pw_args; C2 = 0 / x2 = NULL otherwise); dbias[m] += sum dy[b][m][p] when non-NULL.
 * *_bstride: elements between batch items (0 = contiguous).  dw row m is stored at row perm[m / blk_rows] * blk_rows + m % blk_rows
 * with row stride ldw (blk_rows = 0: identity) -- lets the stacked x_proj GEMM of the fused SS2D write its per-direction blocks. */
typedef struct {
    const float* dy; int64_t dy_bstride; int M;
    const float* x1; int64_t x1_bstride; int C1;
    const float* x2; int64_t x2_bstride; int C2;
    float* dw; int64_t ldw; int blk_rows; int perm[4];
    float* dbias;
    int B; int L;
} bem_wgrad_args;
int bem_pw_wgrad_f32(const bem_wgrad_args* a, void* stream);
/* The same contract for L % 32 == 0 on the bf16 matrix cores: both operands are loaded in MFMA operand order straight from their
 * channel planes (consecutive pixels per lane), split into three bf16 limbs and multiplied as six exact limb products with f32
 * accumulation (the arithmetic of bem_pw_gemm_x6_f32) -- no LDS transposes.  ws: scratch of at least
 * bem_pw_wgrad_x6_ws_elems(M, C1 + C2, B, L) floats (per-workgroup partial tiles, summed by a second kernel: no float atomics). */
int64_t bem_pw_wgrad_x6_ws_elems(int M, int K, int B, int L);
int bem_pw_wgrad_x6_f32(const bem_wgrad_args* a, float* ws, int64_t ws_elems, void* stream);
/* Weight (+ bias) gradient of the dense convolution of bem_conv2d_f32: dw (Cout,Cin,KH,KW) += dy (*) x, rows gathered on the
 * fly (no im2col tensor); dy (B,Cout,Ho,Wo) contiguous, x (B,Cin,H,W) with batch stride x_bstride (0 = contiguous). */
int bem_conv_wgrad_f32(const float* dy, const float* x, int64_t x_bstride, float* dw, float* dbias, int B, int Cin, int H, int W,
                       int Cout, int KH, int KW, int stride, int pad, void* stream);

/* Backward of bem_ss2d_scan_strided_f32 (same operand layout).  dy0 / dy1 = dL/dy0, dL/dy1; dx0 / dx1 written;
 * dxd0 / dxd1 (B,2,R+2,L) contiguous: zeroed by the call, then accumulated over channels; dAlog (4C) = gradient of A_logs,
 * dDs (4C), ddtw (4,C,R), ddtb (4,C) accumulated. */
int bem_ss2d_scan_bwd_f32(const float* x0, const float* x1, const float* xd0, const float* xd1, const float* dy0, const float* dy1,
                          const float* dtw, const float* dtb, const float* A, const float* Ds, float* dx0, float* dx1, float* dxd0,
                          float* dxd1, float* dAlog, float* dDs, float* ddtw, float* ddtb, int B, int C, int L, int R,
                          int64_t xd0_bstride, int64_t xd1_bstride, void* stream);

/* clip_grad_norm_ + torch.optim.AdamW on one flat parameter buffer.  bem_grad_sumsq_f32: acc[0] = sum g^2 (f64, zeroed by the
 * call).  bem_adamw_step_f32: g *= min(1, max_norm / (sqrt(sumsq) + 1e-6)) when max_norm > 0 (read on the device), then the AdamW
 * update with bias corrections of `step` (>= 1); norm_out (or NULL) receives the unclipped total norm.  hyper (NULL or three floats
 * in device memory: lr, 1 - beta1^t, sqrt(1 - beta2^t)) overrides lr / step with values read by the kernel -- the per-iteration
 * inputs of a captured step. */
int bem_grad_sumsq_f32(const float* g, int64_t n, double* acc, void* stream);
int bem_adamw_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int step, float max_norm, const double* sumsq, float* norm_out, const float* hyper,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage-I training: ConditionGenerator.optimize_parameters,
 * basicsr/models/condition_generator_model.py:176-218.  Everything else of that step runs on the Stage-II training entry points.
 * --------------------------------------------------------------------------------------------- */

/* Writes n (<= 512) 32-bit words from HOST memory into device memory at dst, carried as the arguments of one kernel launch on
 * `stream` (ordered with the launches around it; the host buffer may be reused at once).  The per-iteration inputs of a captured
 * Stage-I step (stream_add of the samplers, hyper of bem_adamw_step_f32, decay_dev below) are refreshed with it before each replay. */
int bem_store_words(void* dst, const void* host_words, int n, void* stream);

/* Threshold-EMA prior of a Bayesian leaf (basicsr/bayesian/conv.py:86-98, linear.py:63-74):
 * prior = decay * prior + (1 - decay) * current, for mu and rho; the caller passes decay = min(layer.decay, (1 + step) / (10 + step)),
 * by value or (decay_dev != NULL) as one float in device memory that the kernel reads -- the form a captured step uses. */
int bem_bnn_prior_ema_f32(float* prior_mu, float* prior_rho, const float* mu, const float* rho, float decay, const float* decay_dev,
                          int64_t n, void* stream);

/* The four Bayesian training steps (prior EMA + eps draw + sample, KL, KL backward, reparameterisation backward) for ALL Bayesian
 * tensors of a net in one launch each.  segs: nseg x 8 64-bit words {mu, rho, dmu, drho pointers; first element in the arenas
 * (multiple of 4); elements n; stream counter; float 1/n in the low half of the last word}; blks: nblk x {int32 segment, int32 first
 * element}: one workgroup = 1024 consecutive elements of one segment.  prior_mu / prior_rho / w / eps / gw are flat arenas indexed by
 * segment offset + element.  sample: prior = decay prior + (1 - decay) param; eps = the Philox draw bem_randn_f32 makes for
 * (seed, stream_base + counter [+ *stream_add]); w = mu + log1p(exp(rho)) eps; gw = 0.  kl / kl_bwd / reparam_bwd: as the per-tensor
 * entry points below, gradients accumulated through the dmu / drho pointers.  Replaces conv.py:84-112 + tools.py:76-84 over a net. */
int bem_bnn_bank_sample_f32(const void* segs, const void* blks, int nblk, float* prior_mu, float* prior_rho, float* w, float* eps, float* gw,
                            float decay, const float* decay_dev, uint64_t seed, uint64_t stream_base, const uint64_t* stream_add, void* stream);
int bem_bnn_bank_kl_f32(const void* segs, const void* blks, int nblk, const float* prior_mu, const float* prior_rho, float* out, void* stream);
int bem_bnn_bank_kl_bwd_f32(const void* segs, const void* blks, int nblk, const float* prior_mu, const float* prior_rho, const float* g,
                            void* stream);
int bem_bnn_bank_reparam_bwd_f32(const void* segs, const void* blks, int nblk, const float* gw, const float* eps, void* stream);

/* out[0] += mean( log sp - log sq + (sq^2 + (mu - prior_mu)^2) / (2 sp^2) - 0.5 ), s = log1p(exp(rho))  (base_layer.py:26-40 kl_div). */
int bem_bnn_kl_f32(const float* mu, const float* rho, const float* prior_mu, const float* prior_rho, int64_t n, float* out, void* stream);

/* gradient of g[0] * (that mean) accumulated into dmu / drho (g: one float on the device, the upstream gradient of the KL term). */
int bem_bnn_kl_bwd_f32(const float* mu, const float* rho, const float* prior_mu, const float* prior_rho, int64_t n, const float* g,
                       float* dmu, float* drho, void* stream);

/* Reparameterisation w = mu + log1p(exp(rho)) * eps (conv.py:100-104): dmu += gw, drho += gw * eps * sigmoid(rho). */
int bem_bnn_reparam_bwd_f32(const float* gw, const float* eps, const float* rho, float* dmu, float* drho, int64_t n, void* stream);

/* Masked-image-modelling mix (basicsr/archs/UNet_arch.py:463-466): out = fea * (1 - w) + token[c] * w, w = mask (B,H,W);
 * backward: dfea = dout * (1 - w), dtoken[c] += sum dout * w. */
int bem_mask_token_f32(const float* fea, const float* mask, const float* token, float* out, int B, int C, int H, int W, void* stream);
int bem_mask_token_bwd_f32(const float* dout, const float* mask, float* dfea, float* dtoken, int B, int C, int H, int W, void* stream);

/* Inverse of bem_space_to_depth_f32 (= backward of PatchMerging's gather, UNet_arch.py:74-78): d4 (B,4C,H/2,W/2) -> dx (B,C,H,W). */
int bem_depth_to_space_f32(const float* d4, float* dx, int B, int C, int H, int W, void* stream);

/* nn.PReLU() with one shared slope (DualUpSample, UNet_arch.py:106,120): forward, and backward dx = dout * (x >= 0 ? 1 : a),
 * dslope[0] += sum dout * x over x < 0. */
int bem_prelu_f32(const float* x, const float* slope, float* out, int64_t n, void* stream);
int bem_prelu_bwd_f32(const float* x, const float* slope, const float* dout, float* dx, float* dslope, int64_t n, void* stream);

/* Adjoint of bem_bilinear_up_f32 (nn.Upsample(scale_factor = s, bilinear, align_corners = False), UNet_arch.py:121-123):
 * dout (B,C,H*s,W*s) -> dx (B,C,H,W), zeroed by the call. */
int bem_bilinear_up_bwd_f32(const float* dout, float* dx, int B, int C, int H, int W, int s, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEM_HIP_H */
