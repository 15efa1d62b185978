// Spatial convolutions: depthwise 3x3 (with fused SiLU / gated-GELU / PostSmooth epilogues) and a
// dense direct convolution (3x3 s1, 4x4 s2, ...) staged through LDS.
#include "bem_common.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------
// depthwise 3x3, pad 1.  grid (ceil(ceil(H/RB)*W4/256), Cout, B); each thread produces a 4 (x) by RB (y)
// block of one channel: RB + 2 input rows are read once each as {left scalar, aligned float4, right scalar}.
// ------------------------------------------------------------------------------------------------
constexpr int RB = 4;

__device__ __forceinline__ void dw_block(const float* __restrict__ plane, int H, int W, int y0, int x0, bool vec,
                                         const float* __restrict__ w9, float (&acc)[RB][4]) {
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w9[i];
#pragma unroll
    for (int ry = -1; ry <= RB; ++ry) {
        const int yy = y0 + ry;
        if (yy < 0 || yy >= H) continue;
        const float* r = plane + (int64_t)yy * W;
        float v[6];
        v[0] = x0 > 0 ? r[x0 - 1] : 0.f;
        if (vec) {
            const float4 c = *reinterpret_cast<const float4*>(r + x0);
            v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[1 + i] = (x0 + i < W) ? r[x0 + i] : 0.f;
        }
        v[5] = (x0 + 4 < W) ? r[x0 + 4] : 0.f;
#pragma unroll
        for (int oy = 0; oy < RB; ++oy) {
            const int dy = ry - oy;            // compile-time after unrolling
            if (dy < -1 || dy > 1) continue;
            const float w0 = wk[(dy + 1) * 3], w1 = wk[(dy + 1) * 3 + 1], w2 = wk[(dy + 1) * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[oy][j] = fmaf(w0, v[j], fmaf(w1, v[j + 1], fmaf(w2, v[j + 2], acc[oy][j])));
        }
    }
}

__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        int64_t w_bs, const float* __restrict__ bias, int64_t b_bs,
                                                        float* __restrict__ out, int Cout, int H, int W, int mode) {
    const int W4 = (W + 3) >> 2, HB = (H + RB - 1) / RB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HB * W4) return;
    const int yb = i / W4, y0 = yb * RB, x0 = (i - yb * W4) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    const int Cin = (mode == 2) ? 2 * Cout : Cout;
    const int64_t HW = (int64_t)H * W;
    const bool vec = (W & 3) == 0;
    const float* wb = w + (int64_t)b * w_bs;
    const float* bb = bias ? bias + (int64_t)b * b_bs : nullptr;
    float a0[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[r][j] = 0.f;
    const float* pl = x + ((int64_t)b * Cin + c) * HW;
    dw_block(pl, H, W, y0, x0, vec, wb + (int64_t)c * 9, a0);
    const float b0 = bb ? bb[c] : 0.f;
    float a1[RB][4];
    float b1 = 0.f;
    if (mode == 2) {
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) a1[r][j] = 0.f;
        dw_block(x + ((int64_t)b * Cin + c + Cout) * HW, H, W, y0, x0, vec, wb + (int64_t)(c + Cout) * 9, a1);
        b1 = bb ? bb[c + Cout] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int y = y0 + r;
        if (y >= H) break;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = a0[r][j] + b0;
            if (mode == 2) o[j] = bem_gelu(v) * (a1[r][j] + b1);
            else if (mode == 1) o[j] = bem_silu(v);
            else if (mode == 3) o[j] = ((x0 + j < W) ? pl[(int64_t)y * W + x0 + j] : 0.f) + fmaxf(v, 0.f);
            else o[j] = v;
        }
        float* op = out + ((int64_t)b * Cout + c) * HW + (int64_t)y * W + x0;
        if (vec) {
            *reinterpret_cast<float4*>(op) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < W) op[j] = o[j];
        }
    }
}

// Fast form for W % 4 == 0, 64 % (W/4) == 0, H % RB == 0 (every plane of a 256x256 image): a wavefront covers whole
// image rows, so the left / right halo columns come from the neighbouring lanes' float4 (DPP wave shifts) instead of
// two more loads per row; rows are read at clamped addresses and the two possibly-outside rows are masked by
// multiplication, so the RB + 2 row loads of a plane issue back to back with no branch in between.
__device__ __forceinline__ float dpp_from_prev_lane(float v) {   // lane l <- lane l - 1 (lane 0 <- 0.f)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {   // lane l <- lane l + 1 (lane 63 <- 0.f)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <bool DPPHALO>
__device__ __forceinline__ void dw_block_fast(const float* __restrict__ plane, int H, int W, int y0, int x0,
                                              const float* __restrict__ w9, float (&acc)[RB][4]) {
    float4 c[RB + 2];
    float hl[RB + 2], hr[RB + 2];       // halo columns x0 - 1, x0 + 4 when they cannot come from the neighbouring lanes
#pragma unroll
    for (int ry = -1; ry <= RB; ++ry) {
        const int yy = min(max(y0 + ry, 0), H - 1);
        const float* r = plane + (int64_t)yy * W;
        c[ry + 1] = *reinterpret_cast<const float4*>(r + x0);
        if (!DPPHALO) {                 // clamped addresses; masked by ml / mr below
            hl[ry + 1] = r[max(x0 - 1, 0)];
            hr[ry + 1] = r[min(x0 + 4, W - 1)];
        }
    }
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w9[i];
    const float mt = y0 > 0 ? 1.f : 0.f, mb = y0 + RB < H ? 1.f : 0.f;
    const float ml = x0 > 0 ? 1.f : 0.f, mr = x0 + 4 < W ? 1.f : 0.f;
#pragma unroll
    for (int ry = -1; ry <= RB; ++ry) {
        float4 q = c[ry + 1];
        if (ry == -1) { q.x *= mt; q.y *= mt; q.z *= mt; q.w *= mt; }
        if (ry == RB) { q.x *= mb; q.y *= mb; q.z *= mb; q.w *= mb; }
        const float v[6] = {(DPPHALO ? dpp_from_prev_lane(q.w) : hl[ry + 1] * ((ry == -1) ? mt : (ry == RB) ? mb : 1.f)) * ml, q.x, q.y, q.z, q.w,
                            (DPPHALO ? dpp_from_next_lane(q.x) : hr[ry + 1] * ((ry == -1) ? mt : (ry == RB) ? mb : 1.f)) * mr};
#pragma unroll
        for (int oy = 0; oy < RB; ++oy) {
            const int dy = ry - oy;
            if (dy < -1 || dy > 1) continue;
            const float w0 = wk[(dy + 1) * 3], w1 = wk[(dy + 1) * 3 + 1], w2 = wk[(dy + 1) * 3 + 2];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[oy][j] = fmaf(w0, v[j], fmaf(w1, v[j + 1], fmaf(w2, v[j + 2], acc[oy][j])));
        }
    }
}

template <int MODE, bool DPPHALO>
__global__ __launch_bounds__(256) void dwconv3x3_fast_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             int64_t w_bs, const float* __restrict__ bias, int64_t b_bs,
                                                             float* __restrict__ out, int Cout, int H, int W) {
    const int W4 = W >> 2, HB = H / RB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HB * W4) return;           // whole rows per wavefront: the lanes that leave are past the last row
    const int yb = i / W4, y0 = yb * RB, x0 = (i - yb * W4) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    const int Cin = (MODE == 2) ? 2 * Cout : Cout;
    const int64_t HW = (int64_t)H * W;
    const float* wb = w + (int64_t)b * w_bs;
    // bias through an always-valid pointer and a multiplicative mask (no branch around the load)
    const float* bb = bias ? bias + (int64_t)b * b_bs : wb;
    const float bm = bias ? 1.f : 0.f;
    const float b0 = bb[bias ? c : 0] * bm;
    const float b1 = (MODE == 2) ? bb[bias ? c + Cout : 0] * bm : 0.f;
    const float* pl = x + ((int64_t)b * Cin + c) * HW;
    float a0[RB][4], a1[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[r][j] = a1[r][j] = 0.f;
    dw_block_fast<DPPHALO>(pl, H, W, y0, x0, wb + (int64_t)c * 9, a0);
    if (MODE == 2) dw_block_fast<DPPHALO>(x + ((int64_t)b * Cin + c + Cout) * HW, H, W, y0, x0, wb + (int64_t)(c + Cout) * 9, a1);
    float4 self[RB];
    if (MODE == 3) {
#pragma unroll
        for (int r = 0; r < RB; ++r) self[r] = *reinterpret_cast<const float4*>(pl + (int64_t)(y0 + r) * W + x0);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float o[4];
        const float sv[4] = {MODE == 3 ? self[r].x : 0.f, MODE == 3 ? self[r].y : 0.f, MODE == 3 ? self[r].z : 0.f, MODE == 3 ? self[r].w : 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = a0[r][j] + b0;
            if (MODE == 2) o[j] = bem_gelu_fast(v) * (a1[r][j] + b1);
            else if (MODE == 1) o[j] = bem_silu(v);
            else if (MODE == 3) o[j] = sv[j] + fmaxf(v, 0.f);
            else o[j] = v;
        }
        *reinterpret_cast<float4*>(out + ((int64_t)b * Cout + c) * HW + (int64_t)(y0 + r) * W + x0) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// dense direct conv.  Workgroup = 256 threads = 8 x 32 output pixels, COB output channels per
// workgroup, input channels streamed through LDS in chunks of CIB (patch + weights).
// ------------------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 32, COB = 16, CIB = 8;

template <int KH, int KW, int S>
__global__ __launch_bounds__(256) void conv2d_kernel(const float* __restrict__ x, int64_t x_bs,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ res1, const float* __restrict__ res2,
                                                     float* __restrict__ out, int Cin, int H, int W, int Cout,
                                                     int Ho, int Wo, int pad, int relu, int tilesX) {
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW;
    constexpr int PWP = PW + 1;                       // +1 pad: rows of a patch start on different banks
    __shared__ float patch[CIB][PH][PWP];
    __shared__ __attribute__((aligned(16))) float wl[CIB][KH * KW][COB];
    const int tile = blockIdx.x;
    const int ty0 = (tile / tilesX) * TH, tx0 = (tile % tilesX) * TW;
    const int co0 = blockIdx.y * COB;
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int oy = ty0 + ty, ox = tx0 + tx;
    const int iy0 = ty0 * S - pad, ix0 = tx0 * S - pad;
    const float* xb = x + (int64_t)b * x_bs;
    float acc[COB];
#pragma unroll
    for (int i = 0; i < COB; ++i) acc[i] = 0.f;
    for (int ci0 = 0; ci0 < Cin; ci0 += CIB) {
        __syncthreads();
        for (int i = threadIdx.x; i < CIB * PH * PW; i += 256) {
            const int cc = i / (PH * PW), rem = i - cc * (PH * PW);
            const int py = rem / PW, px = rem - py * PW;
            const int ci = ci0 + cc, iy = iy0 + py, ix = ix0 + px;
            float v = 0.f;
            if (ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) v = xb[((int64_t)ci * H + iy) * W + ix];
            patch[cc][py][px] = v;
        }
        for (int i = threadIdx.x; i < CIB * KH * KW * COB; i += 256) {
            const int o = i % COB, t = (i / COB) % (KH * KW), cc = i / (COB * KH * KW);
            const int ci = ci0 + cc, co = co0 + o;
            wl[cc][t][o] = (ci < Cin && co < Cout) ? w[(((int64_t)co * Cin + ci) * KH * KW) + t] : 0.f;
        }
        __syncthreads();
#pragma unroll 2
        for (int cc = 0; cc < CIB; ++cc) {
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
                for (int kx = 0; kx < KW; ++kx) {
                    const float v = patch[cc][ty * S + ky][tx * S + kx];
                    const float4* wq = reinterpret_cast<const float4*>(&wl[cc][ky * KW + kx][0]);
#pragma unroll
                    for (int q = 0; q < COB / 4; ++q) {
                        const float4 ww = wq[q];
                        acc[4 * q] = fmaf(ww.x, v, acc[4 * q]);
                        acc[4 * q + 1] = fmaf(ww.y, v, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(ww.z, v, acc[4 * q + 2]);
                        acc[4 * q + 3] = fmaf(ww.w, v, acc[4 * q + 3]);
                    }
                }
            }
        }
    }
    if (oy >= Ho || ox >= Wo) return;
#pragma unroll
    for (int o = 0; o < COB; ++o) {
        const int co = co0 + o;
        if (co >= Cout) break;
        float v = acc[o] + (bias ? bias[co] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        const int64_t idx = (((int64_t)b * Cout + co) * Ho + oy) * Wo + ox;
        if (res1) v += res1[idx];
        if (res2) v += res2[idx];
        out[idx] = v;
    }
}


// ------------------------------------------------------------------------------------------------
// dense conv as an implicit GEMM on the f32 matrix cores.
//   M = Cout (all M-tiles in every wave), N = output pixels, K = Cin*KH*KW in the weight tensor's own
//   (ci, ky, kx) order, so the packed 1x1 operand layout (bem_pack_pw_weight_f32 on the (Cout, Cin*KH*KW) view)
//   is the A operand.  One workgroup = 8 x 32 output pixels: wave w owns rows 2w, 2w+1 as two N-tiles.
//   Input channels are streamed through LDS in chunks of CIB; a small LDS table maps k -> patch offset, so the
//   B operand of k-step s is patch[offs[2s + lane/32] + row*S*PWP + (lane%32)*S].
//   The k loop is branch-free (zero weights beyond K).
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS offset of implicit-GEMM row k = (channel-in-chunk, ky, kx) inside the staged patch: a compile-time constant in the
// unrolled k loops (the two halves of a wavefront take k = 2 s and 2 s + 1), so no table lookup sits in front of the reads.
template <int KH, int KW, int PH, int PWP>
__device__ __forceinline__ constexpr int patch_off(int k) {
    return (k / (KH * KW)) * PH * PWP + ((k % (KH * KW)) / KW) * PWP + (k % (KH * KW)) % KW;
}

// epilogue: out = relu?(acc + bias) + res1 + res2.  Optional operands are read through an always-valid pointer (the
// packed weights, finite) at index 0 and masked by multiplication, rows / pixels past the edge at clamped indices, so
// the loads carry no branches; only the stores are predicated.
template <int MTW>
__device__ __forceinline__ void conv_mfma_epilogue(f32x16 (&acc)[MTW][2], const float* __restrict__ Wp, const float* __restrict__ bias,
                                                   const float* __restrict__ res1, const float* __restrict__ res2,
                                                   float* __restrict__ out, int b, int Cout, int Ho, int Wo, int ty0, int tx0,
                                                   int wave, int half, int j, int relu) {
    const int ox = tx0 + j, oxc = min(ox, Wo - 1);
    const float* bp = bias ? bias : Wp;
    const float* p1 = res1 ? res1 : Wp;
    const float* p2 = res2 ? res2 : Wp;
    const float bmk = bias ? 1.f : 0.f, m1 = res1 ? 1.f : 0.f, m2 = res2 ? 1.f : 0.f;
    const int64_t k0 = bias ? -1 : 0, k1 = res1 ? -1 : 0, k2 = res2 ? -1 : 0;     // index masks
    const float lo = relu ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int oy = ty0 + 2 * wave + t, oyc = min(oy, Ho - 1);
        const bool pix = oy < Ho && ox < Wo;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            constexpr int RG = 8;                          // rows per load batch (register budget)
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += RG) {
                float r1[RG], r2[RG], bv[RG];
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const int r = r0 + q;
                    const int co = min(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, Cout - 1);
                    const int64_t idx = (((int64_t)b * Cout + co) * Ho + oyc) * Wo + oxc;
                    bv[q] = bp[co & k0] * bmk;
                    r1[q] = p1[idx & k1] * m1;
                    r2[q] = p2[idx & k2] * m2;
                }
#pragma unroll
                for (int q = 0; q < RG; ++q) {
                    const int r = r0 + q;
                    const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (pix && co < Cout)
                        out[(((int64_t)b * Cout + co) * Ho + oy) * Wo + ox] = fmaxf(acc[m][t][r] + bv[q], lo) + r1[q] + r2[q];
                }
            }
        }
    }
}

// second launch bound = wavefronts per SIMD the register allocation must leave room for: without it the compiler spreads a
// 32-accumulator kernel over 190 registers and two workgroups fill a CU
template <int KH, int KW, int S, int MTW>
__global__ __launch_bounds__(256, MTW == 1 ? 4 : MTW == 2 ? 3 : MTW == 3 ? 2 : 1) void conv2d_mfma_kernel(const float* __restrict__ x, int64_t x_bs,
                                                          const float* __restrict__ Wp, const float* __restrict__ bias,
                                                          const float* __restrict__ res1, const float* __restrict__ res2,
                                                          float* __restrict__ out, int Cin, int H, int W, int Cout,
                                                          int Ho, int Wo, int pad, int relu, int tilesX, int KS) {
    constexpr int KK = KH * KW;
    constexpr int KC = CIB * KK;                          // k values per chunk (even)
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, PWP = PW + 1;
    __shared__ float patch[CIB * PH * PWP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty0 = (tile / tilesX) * TH, tx0 = (tile % tilesX) * TW;
    const int iy0 = ty0 * S - pad, ix0 = tx0 * S - pad;
    const float* xb = x + (int64_t)b * x_bs;
    f32x16 acc[MTW][2];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    const float* wl = Wp + lane;
    const int64_t mts = (int64_t)KS * 64;
    const int rowoff0 = (2 * wave) * S * PWP + j * S, rowoff1 = rowoff0 + S * PWP;

    for (int ci0 = 0; ci0 < Cin; ci0 += CIB) {
        __syncthreads();                                  // previous chunk fully consumed
        {
            constexpr int UB = 8;
            constexpr int TOT = CIB * PH * PW;
            for (int base = 0; base < TOT; base += 256 * UB) {
                float v[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = base + u * 256 + threadIdx.x;
                    const int cc = i / (PH * PW), rem = i - cc * (PH * PW);
                    const int py = rem / PW, px = rem - py * PW;
                    const int ci = ci0 + cc, iy = iy0 + py, ix = ix0 + px;
                    v[u] = 0.f;
                    if (i < TOT && ci < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W) v[u] = xb[((int64_t)ci * H + iy) * W + ix];
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = base + u * 256 + threadIdx.x;
                    if (i < TOT) {
                        const int cc = i / (PH * PW), rem = i - cc * (PH * PW);
                        const int py = rem / PW, px = rem - py * PW;
                        patch[cc * PH * PWP + py * PWP + px] = v[u];
                    }
                }
            }
        }
        __syncthreads();
        const int sg0 = (ci0 * KK) >> 1;                  // global k-step of this chunk's first step (CIB*KK is even)
        constexpr int NS = KC / 2;                        // k-steps per chunk
        constexpr int PB = 4;
        static_assert(NS % PB == 0, "chunk k-steps must be a multiple of the operand batch");
        float an[MTW][PB];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int u = 0; u < PB; ++u)      // clamped address, masked by multiplication (a uniform select becomes a branch + vmcnt(0))
                an[m][u] = wl[m * mts + (int64_t)min(sg0 + u, KS - 1) * 64] * ((sg0 + u < KS) ? 1.f : 0.f);
        for (int s0 = 0; s0 < NS; s0 += PB) {
            float ac[MTW][PB];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int u = 0; u < PB; ++u) {
                    ac[m][u] = an[m][u];
                    const int sn = sg0 + s0 + PB + u;
                    an[m][u] = wl[m * mts + (int64_t)min(sn, KS - 1) * 64] * ((s0 + PB + u < NS && sn < KS) ? 1.f : 0.f);
                }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int k = 2 * (s0 + u) + half, kc = k / KK, kt = k - kc * KK;      // runtime here: unrolling all k-steps of the wide forms overflows the instruction cache
                const int o = kc * PH * PWP + (kt / KW) * PWP + (kt % KW);
                const float v0 = patch[o + rowoff0], v1 = patch[o + rowoff1];
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m][u], v0, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m][u], v1, acc[m][1], 0, 0, 0);
                }
            }
        }
    }
    conv_mfma_epilogue<MTW>(acc, Wp, bias, res1, res2, out, b, Cout, Ho, Wo, ty0, tx0, wave, half, j, relu);
}

// Software-pipelined form of conv2d_mfma_kernel (3x3 stride 1, <= 2 M-tiles).  Both MFMA operands of a channel chunk
// come from LDS (the input patch and that chunk's slice of the packed weights), so the MFMA loop waits on LDS only and
// the global loads of chunk c + 1 -- issued before the loop of chunk c, written to the other LDS buffer after it -- stay
// in flight behind the matrix pipe (vmcnt retires in order: a per-k-step weight load from global would drain them).
// The per-thread staging descriptors (global offset, LDS offset, in-image mask) are chunk-independent and computed once.
template <int KH, int KW, int S, int MTW>
__global__ __launch_bounds__(256, MTW == 1 ? 3 : 2) void conv2d_mfma_pipe_kernel(const float* __restrict__ x, int64_t x_bs,
                                                               const float* __restrict__ Wp, const float* __restrict__ bias,
                                                               const float* __restrict__ res1, const float* __restrict__ res2,
                                                               float* __restrict__ out, int Cin, int H, int W, int Cout,
                                                               int Ho, int Wo, int pad, int relu, int tilesX, int KS) {
    constexpr int KK = KH * KW;
    constexpr int KC = CIB * KK;
    constexpr int NS = KC / 2;                            // k-steps per chunk
    constexpr int PH = (TH - 1) * S + KH, PW = (TW - 1) * S + KW, PWP = PW + 1;
    constexpr int PSZ = CIB * PH * PWP;
    constexpr int TOT = CIB * PH * PW;
    constexpr int NU = (TOT + 255) / 256;
    constexpr int WSZ = NS * 64;                          // packed weight floats per chunk and M-tile
    static_assert(WSZ % 256 == 0, "weight slice must split evenly over the workgroup");
    constexpr int NWU = WSZ / 256;
    __shared__ float patch[2][PSZ + 1];                   // [PSZ] = write-only slot of the threads past the patch
    __shared__ float wsm[2][MTW][WSZ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty0 = (tile / tilesX) * TH, tx0 = (tile % tilesX) * TW;
    const int iy0 = ty0 * S - pad, ix0 = tx0 * S - pad;
    const float* xb = x + (int64_t)b * x_bs;
    const int HW = H * W;
    int goff[NU], lofc[NU];                                // lofc = LDS offset | channel-in-chunk << 20
    float gm[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = u * 256 + threadIdx.x;
        const int cc = i / (PH * PW), rem = i - cc * (PH * PW);
        const int py = rem / PW, px = rem - py * PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool in = i < TOT && iy >= 0 && iy < H && ix >= 0 && ix < W;
        goff[u] = in ? iy * W + ix : 0;
        gm[u] = in ? 1.f : 0.f;
        lofc[u] = (i < TOT ? cc * PH * PWP + py * PWP + px : PSZ) | (min(cc, CIB - 1) << 20);
    }
    f32x16 acc[MTW][2];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    const int64_t mts = (int64_t)KS * 64;
    const int wtot = KS * 64;                              // packed floats per M-tile
    const int rowoff0 = (2 * wave) * S * PWP + j * S, rowoff1 = rowoff0 + S * PWP;
    float v[NU], wv[MTW][NWU];
    // fetch only issues the loads -- as inline asm, because the compiler moves ordinary loads of read-only data down to
    // their first use (behind the MFMAs); arrive() is the matching wait and carries every fetched register as an in/out
    // operand so that no consumer can be scheduled ahead of it.
    auto fetch = [&](int ci0) {                            // global -> registers for the chunk starting at channel ci0
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const float* p = xb + (int64_t)min(ci0 + (lofc[u] >> 20), Cin - 1) * HW + goff[u];
            asm volatile("global_load_dword %0, %1, off" : "=v"(v[u]) : "v"(p));
        }
        const int wbase = ((ci0 * KK) >> 1) * 64;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int u = 0; u < NWU; ++u) {
                const float* p = Wp + m * mts + min(wbase + u * 256 + (int)threadIdx.x, wtot - 1);
                asm volatile("global_load_dword %0, %1, off" : "=v"(wv[m][u]) : "v"(p));
            }
    };
    auto arrive = [&]() {
        static_assert(NU == 11 && NWU == 9 && MTW <= 2, "operand lists below are written for the 3x3 stride-1 chunk");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]));
#pragma unroll
        for (int m = 0; m < MTW; ++m)
            asm volatile("" : "+v"(wv[m][0]), "+v"(wv[m][1]), "+v"(wv[m][2]), "+v"(wv[m][3]), "+v"(wv[m][4]), "+v"(wv[m][5]),
                              "+v"(wv[m][6]), "+v"(wv[m][7]), "+v"(wv[m][8]));
    };
    auto stash = [&](int bufi, int ci0) {
        const int wbase = ((ci0 * KK) >> 1) * 64;
#pragma unroll
        for (int u = 0; u < NU; ++u) patch[bufi][lofc[u] & 0xfffff] = v[u] * (ci0 + (lofc[u] >> 20) < Cin ? gm[u] : 0.f);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int u = 0; u < NWU; ++u)   // k-steps past KS (Cin not a multiple of the chunk) are stored as zero
                wsm[bufi][m][u * 256 + threadIdx.x] = wv[m][u] * (wbase + u * 256 + (int)threadIdx.x < wtot ? 1.f : 0.f);
    };
    fetch(0);
    arrive();
    stash(0, 0);
    __syncthreads();
    int buf = 0;
    for (int ci0 = 0; ci0 < Cin; ci0 += CIB) {
        // unconditional (the last pass re-reads clamped addresses and stashes into the unused buffer): under a uniform
        // `if` the fetch and the stash become one block behind the MFMAs
        fetch(ci0 + CIB);
        __builtin_amdgcn_sched_barrier(0);
        const float* pb = patch[buf];
#pragma unroll      // fully unrolled: as a separate loop block the optimiser sinks the prefetch loads past it, next to their use
        for (int s = 0; s < NS; ++s) {
            const int o = half ? patch_off<KH, KW, PH, PWP>(2 * s + 1) : patch_off<KH, KW, PH, PWP>(2 * s);
            const float v0 = pb[o + rowoff0], v1 = pb[o + rowoff1];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const float a = wsm[buf][m][s * 64 + lane];
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v1, acc[m][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        arrive();
        stash(buf ^ 1, ci0 + CIB);
        __syncthreads();
        buf ^= 1;
    }
    conv_mfma_epilogue<MTW>(acc, Wp, bias, res1, res2, out, b, Cout, Ho, Wo, ty0, tx0, wave, half, j, relu);
}

}  // namespace

extern "C" int bem_dwconv3x3_f32(const float* x, const float* w, int64_t w_bstride, const float* bias,
                                 int64_t bias_bstride, float* out, int B, int Cout, int H, int W, int mode,
                                 void* stream) {
    BEM_REQUIRE(x && w && out, "dwconv3x3: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && Cout > 0 && Cout <= 65535 && H > 0 && W > 0, "dwconv3x3: bad shape");
    BEM_REQUIRE(mode >= 0 && mode <= 3, "dwconv3x3: mode %d", mode);
    if (B == 0) return BEM_OK;
    dim3 grid(cdiv(cdiv(H, RB) * ((W + 3) / 4), 256), Cout, B);
    hipStream_t s = (hipStream_t)stream;
    // branch-free forms: W % 4 == 0 and H % RB == 0; the halo columns come from the neighbouring lanes (DPP) when a wavefront
    // covers whole image rows (64 % (W / 4) == 0: every plane of a 256x256 image), else from two more clamped loads per row
    const bool fast = W % 4 == 0 && H % RB == 0 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    const bool dpp = fast && 64 % (W / 4) == 0;
#define BEM_DW(MODE) do { if (dpp) dwconv3x3_fast_kernel<MODE, true><<<grid, 256, 0, s>>>(x, w, w_bstride, bias, bias_bstride, out, Cout, H, W); \
                          else dwconv3x3_fast_kernel<MODE, false><<<grid, 256, 0, s>>>(x, w, w_bstride, bias, bias_bstride, out, Cout, H, W); } while (0)
    if (!fast) dwconv3x3_kernel<<<grid, 256, 0, s>>>(x, w, w_bstride, bias, bias_bstride, out, Cout, H, W, mode);
    else if (mode == 0) BEM_DW(0);
    else if (mode == 1) BEM_DW(1);
    else if (mode == 2) BEM_DW(2);
    else BEM_DW(3);
#undef BEM_DW
    return bem_check_launch("dwconv3x3");
}

extern "C" int bem_conv2d_f32(const float* x, int64_t x_bstride, const float* w, const float* bias,
                              const float* res1, const float* res2, float* out, int B, int Cin, int H, int W,
                              int Cout, int KH, int KW, int stride, int pad, int relu, void* stream) {
    BEM_REQUIRE(x && w && out, "conv2d: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "conv2d: bad shape");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    BEM_REQUIRE(Ho > 0 && Wo > 0, "conv2d: empty output");
    if (B == 0) return BEM_OK;
    const int tilesX = cdiv(Wo, TW), tilesY = cdiv(Ho, TH);
    dim3 grid(tilesX * tilesY, cdiv(Cout, COB), B);
    hipStream_t s = (hipStream_t)stream;
    if (KH == 3 && KW == 3 && stride == 1)
        conv2d_kernel<3, 3, 1><<<grid, 256, 0, s>>>(x, x_bstride, w, bias, res1, res2, out, Cin, H, W, Cout, Ho, Wo, pad, relu, tilesX);
    else if (KH == 4 && KW == 4 && stride == 2)
        conv2d_kernel<4, 4, 2><<<grid, 256, 0, s>>>(x, x_bstride, w, bias, res1, res2, out, Cin, H, W, Cout, Ho, Wo, pad, relu, tilesX);
    else
        BEM_REQUIRE(false, "conv2d: unsupported kernel %dx%d stride %d (have 3x3 s1, 4x4 s2)", KH, KW, stride);
    return bem_check_launch("conv2d");
}

extern "C" int bem_conv2d_mfma_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias,
                                   const float* res1, const float* res2, float* out, int B, int Cin, int H, int W,
                                   int Cout, int KH, int KW, int stride, int pad, int relu, void* stream) {
    BEM_REQUIRE(x && Wp && out, "conv2d_mfma: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && Cin > 0 && Cout > 0 && Cout <= 160 && H > 0 && W > 0, "conv2d_mfma: bad shape (Cout <= 160)");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    BEM_REQUIRE(Ho > 0 && Wo > 0, "conv2d_mfma: empty output");
    if (B == 0) return BEM_OK;
    const int tilesX = cdiv(Wo, TW), tilesY = cdiv(Ho, TH);
    const int KS = cdiv(Cin * KH * KW, 2), MT = cdiv(Cout, 32);
    dim3 grid(tilesX * tilesY, B);
    hipStream_t s = (hipStream_t)stream;
#define BEM_CONV_LAUNCH(KH_, KW_, S_, MTW_) \
    conv2d_mfma_kernel<KH_, KW_, S_, MTW_><<<grid, 256, 0, s>>>(x, x_bstride, Wp, bias, res1, res2, out, Cin, H, W, Cout, Ho, Wo, pad, relu, tilesX, KS)
    if (KH == 3 && KW == 3 && stride == 1) {
        static const bool pipe = !(getenv("BEM_CONV_PIPE") && getenv("BEM_CONV_PIPE")[0] == '0');
        if (MT <= 2 && pipe) {
            if (MT == 1) conv2d_mfma_pipe_kernel<3, 3, 1, 1><<<grid, 256, 0, s>>>(x, x_bstride, Wp, bias, res1, res2, out, Cin, H, W, Cout, Ho, Wo, pad, relu, tilesX, KS);
            else conv2d_mfma_pipe_kernel<3, 3, 1, 2><<<grid, 256, 0, s>>>(x, x_bstride, Wp, bias, res1, res2, out, Cin, H, W, Cout, Ho, Wo, pad, relu, tilesX, KS);
        }
        else if (MT == 1) BEM_CONV_LAUNCH(3, 3, 1, 1);
        else if (MT == 2) BEM_CONV_LAUNCH(3, 3, 1, 2);
        else if (MT == 3) BEM_CONV_LAUNCH(3, 3, 1, 3);
        else BEM_CONV_LAUNCH(3, 3, 1, 5);
    } else if (KH == 4 && KW == 4 && stride == 2) {
        if (MT == 1) BEM_CONV_LAUNCH(4, 4, 2, 1);
        else if (MT == 2) BEM_CONV_LAUNCH(4, 4, 2, 2);
        else if (MT == 3) BEM_CONV_LAUNCH(4, 4, 2, 3);
        else BEM_CONV_LAUNCH(4, 4, 2, 5);
    } else {
        BEM_REQUIRE(false, "conv2d_mfma: unsupported kernel %dx%d stride %d (have 3x3 s1, 4x4 s2)", KH, KW, stride);
    }
#undef BEM_CONV_LAUNCH
    return bem_check_launch("conv2d_mfma");
}
