// Backward kernels of the Stage-II training step (reference: autograd through
// basicsr/archs/DecompDualBranchDDWavelet_arch.py:301-369 driven by basicsr/models/image_enhancer_model.py:165-216).
//   - L1 loss + its gradient                      (basicsr/losses/losses.py L1Loss, reduction mean)
//   - IWT + Hamilton product backward             (QD/model4.py:20-37, QD/quaternion.py:3-17)
//   - LayerNorm2d backward                        (vmamba.py:58-63)
//   - depthwise 3x3 + SiLU / GELU-gate backward   (vmamba.py:124,130-131,507-515,708-710)
//   - PixelUnshuffle(2), channel sums, axpy       (layout / reduction helpers of the chain)
//   - fused AdamW + global-norm clip on one flat parameter buffer (torch.optim.AdamW, clip_grad_norm_)
// All f32, NCHW planar; reductions over pixels go wave (DPP / shuffles) -> LDS -> one float atomic per workgroup.
#include "bem_common.h"
#include <algorithm>

namespace {

#define GRID1D(n) dim3((unsigned)cdiv64((n), 256))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, BEM_WAVE);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, BEM_WAVE);
    return v;
}

// ---------------------------------------------------------------- L1 loss -------------------------------------------
// sum |pred - gt| in f64 (one atomic per workgroup); dpred = sign(pred - gt) * scale (scale = loss_weight / numel).
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ dpred,
                                                 double* __restrict__ acc, int64_t n, float scale, const float* __restrict__ gmul) {
    __shared__ double sh[4];
    double s = 0.0;
    if (gmul) scale *= gmul[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = pred[i] - gt[i];
        s += fabsf(d);
        if (dpred) dpred[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ void l1_final_kernel(const double* __restrict__ acc, float* __restrict__ loss, double inv_n) { if (loss) loss[0] = (float)(acc[0] * inv_n); }

// ---------------------------------------------------------------- IWT + Hamilton backward ---------------------------
__device__ __forceinline__ void iwt4f(float ll, float hl, float lh, float hh, float (&o)[4]) {
    ll /= 2; hl /= 2; lh /= 2; hh /= 2;
    o[0] = ll - hl - lh + hh;
    o[1] = ll - hl + lh - hh;
    o[2] = ll + hl - lh - hh;
    o[3] = ll + hl + lh + hh;
}
__global__ void iwt_hamilton_bwd_kernel(const float* __restrict__ q1w, const float* __restrict__ q2w, const float* __restrict__ dout,
                                        float* __restrict__ d1, float* __restrict__ d2, int h, int w, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((int64_t)w * h));
    const int64_t hw = (int64_t)h * w;
    const int64_t base = (int64_t)b * 16 * hw + (int64_t)y * w + x;
    const float* a = q1w + base;
    const float* c = q2w + base;
    float P[4][4], Q[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        iwt4f(a[(int64_t)k * hw], a[(int64_t)(4 + k) * hw], a[(int64_t)(8 + k) * hw], a[(int64_t)(12 + k) * hw], P[k]);
        iwt4f(c[(int64_t)k * hw], c[(int64_t)(4 + k) * hw], c[(int64_t)(8 + k) * hw], c[(int64_t)(12 + k) * hw], Q[k]);
    }
    const int W2 = 2 * w;
    const float* gp = dout + (int64_t)b * 3 * 4 * hw + (int64_t)(2 * y) * W2 + 2 * x;
    float g[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float2 t0 = *reinterpret_cast<const float2*>(gp + (int64_t)k * 4 * hw);
        const float2 t1 = *reinterpret_cast<const float2*>(gp + (int64_t)k * 4 * hw + W2);
        g[k][0] = t0.x; g[k][2] = t0.y; g[k][1] = t1.x; g[k][3] = t1.y;
    }
    float dP[4][4], dQ[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float p0 = P[0][s], p1 = P[1][s], p2 = P[2][s], p3 = P[3][s];
        const float q0 = Q[0][s], q1 = Q[1][s], q2 = Q[2][s], q3 = Q[3][s];
        const float g0 = g[0][s], g1 = g[1][s], g2 = g[2][s];
        dP[0][s] = g0 * q1 + g1 * q2 + g2 * q3;
        dP[1][s] = g0 * q0 - g1 * q3 + g2 * q2;
        dP[2][s] = g0 * q3 + g1 * q0 - g2 * q1;
        dP[3][s] = -g0 * q2 + g1 * q1 + g2 * q0;
        dQ[0][s] = g0 * p1 + g1 * p2 + g2 * p3;
        dQ[1][s] = g0 * p0 + g1 * p3 - g2 * p2;
        dQ[2][s] = -g0 * p3 + g1 * p0 + g2 * p1;
        dQ[3][s] = g0 * p2 - g1 * p1 + g2 * p0;
    }
    float* o1 = d1 + base;
    float* o2 = d2 + base;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        {
            const float* v = dP[k];
            o1[(int64_t)k * hw] = 0.5f * (v[0] + v[1] + v[2] + v[3]);
            o1[(int64_t)(4 + k) * hw] = 0.5f * (-v[0] - v[1] + v[2] + v[3]);
            o1[(int64_t)(8 + k) * hw] = 0.5f * (-v[0] + v[1] - v[2] + v[3]);
            o1[(int64_t)(12 + k) * hw] = 0.5f * (v[0] - v[1] - v[2] + v[3]);
        }
        {
            const float* v = dQ[k];
            o2[(int64_t)k * hw] = 0.5f * (v[0] + v[1] + v[2] + v[3]);
            o2[(int64_t)(4 + k) * hw] = 0.5f * (-v[0] - v[1] + v[2] + v[3]);
            o2[(int64_t)(8 + k) * hw] = 0.5f * (-v[0] + v[1] - v[2] + v[3]);
            o2[(int64_t)(12 + k) * hw] = 0.5f * (v[0] - v[1] - v[2] + v[3]);
        }
    }
}

// ---------------------------------------------------------------- layout / reductions -------------------------------
// nn.PixelUnshuffle(2): (B,C,2H,2W) -> (B,4C,H,W), out[c*4 + i*2 + j][y][x] = in[c][2y+i][2x+j]  (inverse of bem_pixel_shuffle2_f32)
__global__ void pixel_unshuffle2_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % W), y = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
    const int64_t hw = (int64_t)H * W;
    const float* p = x + ((int64_t)b * C + c) * 4 * hw + (int64_t)(2 * y) * (2 * W) + 2 * xx;
    const float2 t0 = *reinterpret_cast<const float2*>(p), t1 = *reinterpret_cast<const float2*>(p + 2 * W);
    float* o = out + ((int64_t)b * 4 * C + 4 * c) * hw + (int64_t)y * W + xx;
    o[0] = t0.x; o[hw] = t0.y; o[2 * hw] = t1.x; o[3 * hw] = t1.y;
}

// out[c] += sum_{b, p} x[b][c][p]   grid (nsplit, C)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int64_t L) {
    __shared__ float sh[4];
    const int c = blockIdx.y;
    float s = 0.f;
    const int64_t n = (int64_t)B * L;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / L, p = i - b * L;
        s += x[(b * C + c) * L + p];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + c, sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t n, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fmaf(alpha, b[i], a[i]);
}

// ---------------------------------------------------------------- LayerNorm2d backward ------------------------------
// One thread per pixel (lanes = consecutive pixels of one image: every channel plane access is a coalesced row segment),
// three sweeps over the C planes of the tile (they stay in L1 / L2): mean; variance + the two projections of g = dn * gamma;
// dx.  dgamma / dbeta: wave sums per channel -> LDS (C <= 1024) -> one atomic per channel and workgroup.
//   x = x1 (+ x2), n = (x - mu) rstd gamma + beta, dx = dres + rstd (g - mean(g) - xhat mean(g xhat))
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ dn,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ n_out,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int64_t L, int64_t tiles_per_img) {
    extern __shared__ float sm[];          // [2*C] dgamma, dbeta partials
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    const int64_t b = blockIdx.x / tiles_per_img, tile = blockIdx.x % tiles_per_img;
    const int64_t p = tile * blockDim.x + threadIdx.x;
    const bool ok = p < L;
    const int64_t pc = ok ? p : L - 1;     // clamped: every lane loads, the outside ones are masked out of the sums / stores
    const int64_t base = b * C * L + pc;
    const float invC = 1.f / (float)C;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f);
    mu *= invC;
    float var = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu;
        const float g = dn[base + c * L] * gamma[c];
        var = fmaf(d, d, var);
        s1 += g;
        s2 = fmaf(g, d, s2);
    }
    const float rstd = rsqrtf(var * invC + eps);
    s1 *= invC;
    s2 *= invC * rstd;                      // mean(g * xhat)
    const int lane = threadIdx.x & 63;
    for (int c = 0; c < C; ++c) {
        const float xh = (x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu) * rstd;
        const float d = dn[base + c * L];
        const float g = d * gamma[c];
        float r = rstd * (g - s1 - xh * s2);
        if (dres) r += dres[base + c * L];
        if (ok) {
            dx[base + c * L] = r;
            if (n_out) n_out[base + c * L] = fmaf(xh, gamma[c], beta[c]);
        }
        const float dg = wave_sum(ok ? d * xh : 0.f), db = wave_sum(ok ? d : 0.f);
        if (lane == 0) { atomicAdd(&sm[c], dg); atomicAdd(&sm[C + c], db); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x) {
        atomicAdd(dgamma + i, sm[i]);
        atomicAdd(dbeta + i, sm[C + i]);
    }
}

// Planes of a few pixels under many channels (the Stage-I bottleneck: 2x2 .. 4x4 planes, C > 160): the pixel-per-thread form above leaves
// 4 .. 16 lanes walking C channels one after the other.  Here a workgroup owns ONE pixel and its threads stride over the channels; the
// three statistics are two workgroup reductions, dgamma / dbeta go to memory as one atomic per (pixel, channel).
__global__ __launch_bounds__(256) void ln_bwd_chan_kernel(const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ dn,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ n_out,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int64_t L) {
    __shared__ float sh[3][4];
    const int64_t b = blockIdx.x / L, p = blockIdx.x % L;
    const int64_t base = b * C * L + p;
    const float invC = 1.f / (float)C;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float a0 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) a0 += x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f);
    a0 = wave_sum(a0);
    if (lane == 0) sh[0][wv] = a0;
    __syncthreads();
    const float mu = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) * invC;
    __syncthreads();
    float var = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float d = x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu;
        const float g = dn[base + c * L] * gamma[c];
        var = fmaf(d, d, var);
        s1 += g;
        s2 = fmaf(g, d, s2);
    }
    var = wave_sum(var); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { sh[0][wv] = var; sh[1][wv] = s1; sh[2][wv] = s2; }
    __syncthreads();
    var = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    s1 = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) * invC;
    const float rstd = rsqrtf(var * invC + eps);
    s2 = (sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]) * invC * rstd;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float xh = (x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu) * rstd;
        const float d = dn[base + c * L];
        float r = rstd * (d * gamma[c] - s1 - xh * s2);
        if (dres) r += dres[base + c * L];
        dx[base + c * L] = r;
        if (n_out) n_out[base + c * L] = fmaf(xh, gamma[c], beta[c]);
        atomicAdd(dgamma + c, d * xh);
        atomicAdd(dbeta + c, d);
    }
}

// Faster form for C <= 160: the four wavefronts of a workgroup split the channels of the same 64 pixels (CPT = ceil(C / 4) each),
// x and dn are read ONCE into registers (all loads of a tile issue back to back), the per-pixel statistics are combined through
// LDS (two barriers per tile), dgamma / dbeta accumulate in registers over the workgroup's tiles (one atomic per channel at the end).
template <int CPT, int NWV, bool X2, bool DRES, bool NOUT>
__global__ __launch_bounds__(64 * NWV) void ln_bwd_split_kernel(const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ dn,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ n_out,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int64_t L,
                                                                int64_t tiles_per_img, int64_t total_tiles, int tiles_per_wg) {
    __shared__ float red[4][NWV][64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: channel indices, gamma / beta stay scalar
    const int c0 = q * CPT;
    const float invC = 1.f / (float)C;
    float ag[CPT], ab[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) ag[j] = ab[j] = 0.f;
    const int64_t t_end = min((int64_t)(blockIdx.x + 1) * tiles_per_wg, total_tiles);
    for (int64_t t = (int64_t)blockIdx.x * tiles_per_wg; t < t_end; ++t) {
        const int64_t b = t / tiles_per_img, tile = t - b * tiles_per_img;
        const int64_t p = tile * 64 + lane;
        const bool ok = p < L;
        const int64_t base = b * C * L + (ok ? p : L - 1);
        float xv[CPT], dv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) xv[j] = x1[base + (int64_t)min(c0 + j, C - 1) * L];
        if (X2) {                                           // compile-time variants: no branch sits between the loads of a tile
#pragma unroll
            for (int j = 0; j < CPT; ++j) xv[j] += x2[base + (int64_t)min(c0 + j, C - 1) * L];
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) dv[j] = dn[base + (int64_t)min(c0 + j, C - 1) * L];
        float rv[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) rv[j] = DRES ? dres[base + (int64_t)min(c0 + j, C - 1) * L] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) s += (c0 + j < C) ? xv[j] : 0.f;
        red[0][q][lane] = s;
        __syncthreads();
        float mu = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) mu += red[0][w][lane];
        mu *= invC;
        float var = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const bool live = c0 + j < C;
            const float d = live ? xv[j] - mu : 0.f;
            const float g = live ? dv[j] * gamma[min(c0 + j, C - 1)] : 0.f;
            xv[j] = d;
            var = fmaf(d, d, var);
            s1 += g;
            s2 = fmaf(g, d, s2);
        }
        red[1][q][lane] = var; red[2][q][lane] = s1; red[3][q][lane] = s2;
        __syncthreads();
        var = s1 = s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) { var += red[1][w][lane]; s1 += red[2][w][lane]; s2 += red[3][w][lane]; }
        const float rstd = rsqrtf(var * invC + eps);
        s1 *= invC;
        s2 *= invC * rstd;
        const float okm = ok ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = min(c0 + j, C - 1);
            const float gm = gamma[c];
            const float xh = xv[j] * rstd;
            float r = rstd * (dv[j] * gm - s1 - xh * s2);
            const int64_t idx = base + (int64_t)c * L;
            r += rv[j];
            if (ok && c0 + j < C) {
                dx[idx] = r;
                if (NOUT) n_out[idx] = fmaf(xh, gm, beta[c]);
            }
            ag[j] = fmaf(dv[j] * okm, xh, ag[j]);
            ab[j] = fmaf(dv[j], okm, ab[j]);
        }
    }
    // dgamma / dbeta of this workgroup: wave sums -> LDS (channel order) -> two contiguous atomic wave-instructions per 64 channels
    // (single-lane atomics from thousands of wavefronts onto the same few addresses serialise at the memory side)
    __syncthreads();
    float* gsum = &red[0][0][0];                        // [2][NWV * CPT] <= 4 * NWV * 64 floats
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float a = wave_sum(ag[j]), b2 = wave_sum(ab[j]);
        if (lane == 0) { gsum[c0 + j] = a; gsum[NWV * CPT + c0 + j] = b2; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64 * NWV) {
        atomicAdd(dgamma + c, gsum[c]);
        atomicAdd(dbeta + c, gsum[NWV * CPT + c]);
    }
}

// LayerNorm2d forward alone (the normalised tensor of a weight-gradient GEMM when no backward pass through the norm is needed)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float* __restrict__ n_out, int C, int64_t L,
                                                     int64_t tiles_per_img) {
    const int64_t b = blockIdx.x / tiles_per_img, tile = blockIdx.x % tiles_per_img;
    const int64_t p = tile * blockDim.x + threadIdx.x;
    if (p >= L) return;
    const int64_t base = b * C * L + p;
    const float invC = 1.f / (float)C;
    float mu = 0.f;
    for (int c = 0; c < C; ++c) mu += x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f);
    mu *= invC;
    float var = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu;
        var = fmaf(d, d, var);
    }
    const float rstd = rsqrtf(var * invC + eps);
    for (int c = 0; c < C; ++c)
        n_out[base + c * L] = fmaf((x1[base + c * L] + (x2 ? x2[base + c * L] : 0.f) - mu) * rstd, gamma[c], beta[c]);
}

// ---------------------------------------------------------------- depthwise 3x3 + activation backward ---------------
// Forward (bem_dwconv3x3_f32): MODE 1  out[c] = SiLU(dw(t[c]) + b[c]);  MODE 2  out[c] = GELU(dw(t[c]) + b[c]) * (dw(t[c+Hd]) + b[c+Hd]).
// This kernel recomputes the pre-activation from t, writes dpre = dL/d(dw(t) + b) (C planes in MODE 1, 2*Hd in MODE 2) and
// accumulates the depthwise weight / bias gradients  dW[c][ky][kx] += sum_p dpre[p] t[p + (ky-1, kx-1)],  db[c] += sum_p dpre[p].
// The input gradient is then the same depthwise convolution with the flipped kernel applied to dpre (bem_dwconv3x3_f32, mode 0).
// One thread per RBB x 4 block of one channel (pair), the (RBB + 2) x 6 window of t held in registers.
constexpr int RBB = 2;

__device__ __forceinline__ void dw_window(const float* __restrict__ plane, int H, int W, int y0, int x0, bool vec, float (&v)[RBB + 2][6]) {
#pragma unroll
    for (int ry = -1; ry <= RBB; ++ry) {
        const int yy = y0 + ry;
        const bool rowok = yy >= 0 && yy < H;
        const float* r = plane + (int64_t)(rowok ? yy : 0) * W;
        const float m = rowok ? 1.f : 0.f;
        v[ry + 1][0] = r[max(x0 - 1, 0)] * (x0 > 0 ? m : 0.f);               // clamped address + mask: no branch next to the load
        if (vec) {
            const float4 c = *reinterpret_cast<const float4*>(r + x0);
            v[ry + 1][1] = c.x * m; v[ry + 1][2] = c.y * m; v[ry + 1][3] = c.z * m; v[ry + 1][4] = c.w * m;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[ry + 1][1 + i] = ((x0 + i < W) ? r[x0 + i] : 0.f) * m;
        }
        v[ry + 1][5] = r[min(x0 + 4, W - 1)] * (x0 + 4 < W ? m : 0.f);
    }
}
__device__ __forceinline__ void dw_apply(const float (&v)[RBB + 2][6], const float* __restrict__ w9, float bias, float (&pre)[RBB][4]) {
    float wk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) wk[i] = w9[i];
#pragma unroll
    for (int oy = 0; oy < RBB; ++oy)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = bias;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) a = fmaf(wk[ky * 3 + kx], v[oy + ky][j + kx], a);
            pre[oy][j] = a;
        }
}
// s[0..8] += sum dpre * window, s[9] += sum dpre   (masked positions carry dpre = 0)
__device__ __forceinline__ void dw_corr(const float (&v)[RBB + 2][6], const float (&d)[RBB][4], float (&s)[10]) {
#pragma unroll
    for (int oy = 0; oy < RBB; ++oy)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) s[ky * 3 + kx] = fmaf(d[oy][j], v[oy + ky][j + kx], s[ky * 3 + kx]);
            s[9] += d[oy][j];
        }
}

template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void dwact_bwd_kernel(const float* __restrict__ t, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ dout, float* __restrict__ dpre, float* __restrict__ dw,
                                                        float* __restrict__ dbias, int Cout, int H, int W) {
    __shared__ float sh[4][20];
    const int W4 = (W + 3) >> 2, HB = (H + RBB - 1) / RBB;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < HB * W4;
    const int ii = active ? i : 0;
    const int yb = ii / W4, y0 = yb * RBB, x0 = (ii - yb * W4) * 4;
    const int c = blockIdx.y, b = blockIdx.z;
    const int Cin = (MODE == 2) ? 2 * Cout : Cout;
    const int64_t HW = (int64_t)H * W;
    constexpr bool vec = VEC;                                   // W % 4 == 0, a compile-time variant: no per-row branches around the loads
    float s0[10], s1[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) s0[k] = s1[k] = 0.f;
    {
        float v0[RBB + 2][6], pre0[RBB][4], g[RBB][4], d0[RBB][4];
        dw_window(t + ((int64_t)b * Cin + c) * HW, H, W, y0, x0, vec, v0);
        dw_apply(v0, w + (int64_t)c * 9, bias ? bias[c] : 0.f, pre0);
        const float* gp = dout + ((int64_t)b * Cout + c) * HW;
#pragma unroll
        for (int oy = 0; oy < RBB; ++oy) {
            if (vec) {                                          // W % 4 == 0: the four columns are in range together, one 16-byte load
                const int yc = min(y0 + oy, H - 1);
                const float4 q = *reinterpret_cast<const float4*>(gp + (int64_t)yc * W + x0);
                const float mk = (active && y0 + oy < H) ? 1.f : 0.f;
                g[oy][0] = q.x * mk; g[oy][1] = q.y * mk; g[oy][2] = q.z * mk; g[oy][3] = q.w * mk;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = active && (y0 + oy < H) && (x0 + j < W);
                    g[oy][j] = ok ? gp[(int64_t)(y0 + oy) * W + x0 + j] : 0.f;
                }
            }
        }
        if (MODE == 2) {
            float v1[RBB + 2][6], pre1[RBB][4], d1[RBB][4];
            dw_window(t + ((int64_t)b * Cin + c + Cout) * HW, H, W, y0, x0, vec, v1);
            dw_apply(v1, w + (int64_t)(c + Cout) * 9, bias ? bias[c + Cout] : 0.f, pre1);
#pragma unroll
            for (int oy = 0; oy < RBB; ++oy)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h1 = pre0[oy][j], h2 = pre1[oy][j];
                    // erf by Abramowitz-Stegun 7.1.26 on the hardware rcp / exp (|error| <= 1.5e-7, bem_common.h) instead of libm's branchy erff
                    const float cdf = 0.5f * (1.f + bem_erf_fast(h1 * 0.70710678118654752440f));
                    const float pdf = 0.3989422804014327f * bem_fexp(-0.5f * h1 * h1);
                    d0[oy][j] = g[oy][j] * h2 * (cdf + h1 * pdf);
                    d1[oy][j] = g[oy][j] * h1 * cdf;
                }
            dw_corr(v1, d1, s1);
            float* o1 = dpre + ((int64_t)b * Cin + c + Cout) * HW;
#pragma unroll
            for (int oy = 0; oy < RBB; ++oy) {
                if (vec) {
                    if (active && y0 + oy < H) *reinterpret_cast<float4*>(o1 + (int64_t)(y0 + oy) * W + x0) = make_float4(d1[oy][0], d1[oy][1], d1[oy][2], d1[oy][3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (active && y0 + oy < H && x0 + j < W) o1[(int64_t)(y0 + oy) * W + x0 + j] = d1[oy][j];
                }
            }
        } else {
#pragma unroll
            for (int oy = 0; oy < RBB; ++oy)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = pre0[oy][j];
                    if (MODE == 1) {
                        const float sg = 1.f / (1.f + __expf(-z));
                        d0[oy][j] = g[oy][j] * sg * (1.f + z * (1.f - sg));
                    } else {
                        d0[oy][j] = g[oy][j];
                    }
                }
        }
        dw_corr(v0, d0, s0);
        float* o0 = dpre + ((int64_t)b * Cin + c) * HW;
#pragma unroll
        for (int oy = 0; oy < RBB; ++oy) {
            if (vec) {
                if (active && y0 + oy < H) *reinterpret_cast<float4*>(o0 + (int64_t)(y0 + oy) * W + x0) = make_float4(d0[oy][0], d0[oy][1], d0[oy][2], d0[oy][3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (active && y0 + oy < H && x0 + j < W) o0[(int64_t)(y0 + oy) * W + x0 + j] = d0[oy][j];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const float a = wave_sum(s0[k]);
        const float bsum = (MODE == 2) ? wave_sum(s1[k]) : 0.f;
        if (lane == 0) { sh[wave][k] = a; sh[wave][10 + k] = bsum; }
    }
    __syncthreads();
    if (threadIdx.x < 20) {
        const int k = threadIdx.x % 10, second = threadIdx.x / 10;
        if (second && MODE != 2) return;
        const float v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        const int ch = second ? c + Cout : c;
        if (k < 9) atomicAdd(dw + (int64_t)ch * 9 + k, v);
        else if (dbias) atomicAdd(dbias + ch, v);
    }
}

// ---------------------------------------------------------------- optimizer ------------------------------------------
// sum of squares of the flat gradient buffer in f64 (clip_grad_norm_'s total norm, torch/nn/utils/clip_grad.py)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ acc) {
    __shared__ double sh[4];
    double s = 0.0;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s += (double)v * v; }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}

// torch.optim.AdamW (decoupled weight decay, bias-corrected) on a flat buffer, with clip_grad_norm_(max_norm) folded in:
// the clip coefficient min(1, max_norm / (sqrt(sumsq) + 1e-6)) is read from the device (no host round trip) and the clipped
// gradient is written back, as clip_grad_norm_ does in place.  max_norm <= 0: no clipping.
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float max_norm,
                             const double* __restrict__ sumsq, float* __restrict__ norm_out, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }     // per-iteration values of a captured step, kept in HBM
    float coef = 1.f;
    if (max_norm > 0.f) {
        const float tn = (float)sqrt(sumsq[0]);
        coef = fminf(max_norm / (tn + 1e-6f), 1.f);
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = tn;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * coef;
    g[i] = gi;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
}

}  // namespace

extern "C" int bem_l1_loss_f32(const float* pred, const float* gt, float* dpred, float* loss, double* ws, int64_t n, float weight,
                               const float* gmul, void* stream) {
    BEM_REQUIRE(pred && gt && (loss || dpred) && ws && n > 0, "l1_loss: null pointer / empty tensor");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double), s) != hipSuccess) return bem_check_launch("l1_loss memset");
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, 256), 2048);
    l1_kernel<<<grid, 256, 0, s>>>(pred, gt, dpred, ws, n, weight / (float)n, gmul);
    l1_final_kernel<<<1, 1, 0, s>>>(ws, loss, (double)weight / (double)n);
    return bem_check_launch("l1_loss");
}

namespace {
// backward of bem_hamilton_f32: q (B,8,HW) = [p | q], dout (B,3,HW) = d(i, j, k parts of p x q)  ->  dq8 (B,8,HW) = [dp | dq]
__global__ void hamilton_bwd_kernel(const float* __restrict__ q8, const float* __restrict__ dout, float* __restrict__ dq8, int64_t HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pix = i % HW, b = i / HW;
    const float* s = q8 + b * 8 * HW + pix;
    const float p[4] = {s[0], s[HW], s[2 * HW], s[3 * HW]}, q[4] = {s[4 * HW], s[5 * HW], s[6 * HW], s[7 * HW]};
    const float* g = dout + b * 3 * HW + pix;
    const float gi = g[0], gj = g[HW], gk = g[2 * HW];
    float* d = dq8 + b * 8 * HW + pix;
    // i = p0 q1 + p1 q0 + p2 q3 - p3 q2;  j = p0 q2 - p1 q3 + p2 q0 + p3 q1;  k = p0 q3 + p1 q2 - p2 q1 + p3 q0
    d[0] = gi * q[1] + gj * q[2] + gk * q[3];
    d[HW] = gi * q[0] - gj * q[3] + gk * q[2];
    d[2 * HW] = gi * q[3] + gj * q[0] - gk * q[1];
    d[3 * HW] = -gi * q[2] + gj * q[1] + gk * q[0];
    d[4 * HW] = gi * p[1] + gj * p[2] + gk * p[3];
    d[5 * HW] = gi * p[0] + gj * p[3] - gk * p[2];
    d[6 * HW] = -gi * p[3] + gj * p[0] + gk * p[1];
    d[7 * HW] = gi * p[2] - gj * p[1] + gk * p[0];
}
}  // namespace

extern "C" int bem_hamilton_bwd_f32(const float* q8, const float* dout, float* dq8, int B, int H, int W, void* stream) {
    BEM_REQUIRE(q8 && dout && dq8 && B > 0 && H > 0 && W > 0, "hamilton_bwd: bad arguments");
    const int64_t HW = (int64_t)H * W, total = (int64_t)B * HW;
    hamilton_bwd_kernel<<<GRID1D(total), 256, 0, (hipStream_t)stream>>>(q8, dout, dq8, HW, total);
    return bem_check_launch("hamilton_bwd");
}

extern "C" int bem_iwt_hamilton_bwd_f32(const float* q1w, const float* q2w, const float* dout, float* dq1w, float* dq2w, int B, int h, int w,
                                        void* stream) {
    BEM_REQUIRE(q1w && q2w && dout && dq1w && dq2w && B > 0 && h > 0 && w > 0, "iwt_hamilton_bwd: bad arguments");
    const int64_t total = (int64_t)B * h * w;
    iwt_hamilton_bwd_kernel<<<GRID1D(total), 256, 0, (hipStream_t)stream>>>(q1w, q2w, dout, dq1w, dq2w, h, w, total);
    return bem_check_launch("iwt_hamilton_bwd");
}

extern "C" int bem_pixel_unshuffle2_f32(const float* x, float* out, int B, int C, int H, int W, void* stream) {
    BEM_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "pixel_unshuffle2: bad arguments (H, W = output plane size)");
    const int64_t total = (int64_t)B * C * H * W;
    pixel_unshuffle2_kernel<<<GRID1D(total), 256, 0, (hipStream_t)stream>>>(x, out, C, H, W, total);
    return bem_check_launch("pixel_unshuffle2");
}

extern "C" int bem_channel_sum_f32(const float* x, float* out, int B, int C, int64_t L, void* stream) {
    BEM_REQUIRE(x && out && B > 0 && C > 0 && C <= 65535 && L > 0, "channel_sum: bad arguments");
    const unsigned nsplit = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv64((int64_t)B * L, 256 * 16), 2048 / C + 1));
    channel_sum_kernel<<<dim3(nsplit, C), 256, 0, (hipStream_t)stream>>>(x, out, B, C, L);
    return bem_check_launch("channel_sum");
}

extern "C" int bem_add_f32(const float* a, const float* b, float* out, int64_t n, float alpha, void* stream) {
    BEM_REQUIRE(a && b && out && n > 0, "add: bad arguments");
    add_kernel<<<GRID1D(n), 256, 0, (hipStream_t)stream>>>(a, b, out, n, alpha);
    return bem_check_launch("add");
}

extern "C" int bem_ln_bwd_f32(const float* x1, const float* x2, const float* dn, const float* gamma, const float* beta, float eps,
                              const float* dres, float* dx, float* n_out, float* dgamma, float* dbeta, int B, int C, int64_t L, void* stream) {
    BEM_REQUIRE(x1 && dn && gamma && beta && dx && dgamma && dbeta, "ln_bwd: null pointer");
    BEM_REQUIRE(B > 0 && C > 0 && C <= 4096 && L > 0, "ln_bwd: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (C <= 160) {
        const int64_t tpi = cdiv64(L, 64), total = tpi * B;
        const int tpw = (int)std::max<int64_t>(1, cdiv64(total, 1024));
        const unsigned grid = (unsigned)cdiv64(total, tpw);
#define BEM_LNB3(CPT, NWV, A, D, N) ln_bwd_split_kernel<CPT, NWV, A, D, N><<<grid, 64 * NWV, 0, s>>>(x1, x2, dn, gamma, beta, eps, dres, dx, n_out, dgamma, dbeta, C, L, tpi, total, tpw)
#define BEM_LNB(CPT, NWV)                                                                                        \
    do {                                                                                                         \
        const int v = (x2 ? 4 : 0) | (dres ? 2 : 0) | (n_out ? 1 : 0);                                           \
        switch (v) {                                                                                             \
            case 0: BEM_LNB3(CPT, NWV, false, false, false); break;                                              \
            case 1: BEM_LNB3(CPT, NWV, false, false, true); break;                                               \
            case 2: BEM_LNB3(CPT, NWV, false, true, false); break;                                               \
            case 3: BEM_LNB3(CPT, NWV, false, true, true); break;                                                \
            case 4: BEM_LNB3(CPT, NWV, true, false, false); break;                                               \
            case 5: BEM_LNB3(CPT, NWV, true, false, true); break;                                                \
            case 6: BEM_LNB3(CPT, NWV, true, true, false); break;                                                \
            default: BEM_LNB3(CPT, NWV, true, true, true); break;                                                \
        }                                                                                                        \
    } while (0)
        if (C <= 16) BEM_LNB(4, 4);
        else if (C <= 40) BEM_LNB(10, 4);
        else if (C <= 80) BEM_LNB(10, 8);
        else BEM_LNB(20, 8);
#undef BEM_LNB
#undef BEM_LNB3
        return bem_check_launch("ln_bwd");
    }
    if (L <= 16) {
        ln_bwd_chan_kernel<<<dim3((unsigned)(B * L)), 256, 0, s>>>(x1, x2, dn, gamma, beta, eps, dres, dx, n_out, dgamma, dbeta, C, L);
        return bem_check_launch("ln_bwd");
    }
    const int64_t tiles = cdiv64(L, 256);
    BEM_REQUIRE(tiles * B < (1ll << 31), "ln_bwd: grid too large");
    ln_bwd_kernel<<<dim3((unsigned)(tiles * B)), 256, 2 * C * sizeof(float), s>>>(x1, x2, dn, gamma, beta, eps, dres, dx, n_out,
                                                                                 dgamma, dbeta, C, L, tiles);
    return bem_check_launch("ln_bwd");
}

extern "C" int bem_ln_fwd_f32(const float* x1, const float* x2, const float* gamma, const float* beta, float eps, float* n_out, int B, int C,
                              int64_t L, void* stream) {
    BEM_REQUIRE(x1 && gamma && beta && n_out && B > 0 && C > 0 && L > 0, "ln_fwd: bad arguments");
    const int64_t tiles = cdiv64(L, 256);
    BEM_REQUIRE(tiles * B < (1ll << 31), "ln_fwd: grid too large");
    ln_fwd_kernel<<<dim3((unsigned)(tiles * B)), 256, 0, (hipStream_t)stream>>>(x1, x2, gamma, beta, eps, n_out, C, L, tiles);
    return bem_check_launch("ln_fwd");
}

extern "C" int bem_dwact_bwd_f32(const float* t, const float* w, const float* bias, const float* dout, float* dpre, float* dw, float* dbias,
                                 int B, int Cout, int H, int W, int mode, void* stream) {
    BEM_REQUIRE(t && w && dout && dpre && dw, "dwact_bwd: null pointer");
    BEM_REQUIRE(B > 0 && Cout > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2, "dwact_bwd: bad sizes / mode");
    BEM_REQUIRE(Cout <= 65535 && B <= 65535, "dwact_bwd: grid too large");
    BEM_REQUIRE(!bias == !dbias, "dwact_bwd: bias and dbias go together");
    const int W4 = (W + 3) >> 2, HB = (H + RBB - 1) / RBB;
    const dim3 grid((unsigned)cdiv(HB * W4, 256), (unsigned)Cout, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (W & 3) == 0 && ((uintptr_t)t & 15) == 0 && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)dpre & 15) == 0;
#define BEM_DWACT(M)                                                                                             \
    do {                                                                                                         \
        if (vec) dwact_bwd_kernel<M, true><<<grid, 256, 0, s>>>(t, w, bias, dout, dpre, dw, dbias, Cout, H, W);  \
        else dwact_bwd_kernel<M, false><<<grid, 256, 0, s>>>(t, w, bias, dout, dpre, dw, dbias, Cout, H, W);     \
    } while (0)
    if (mode == 2) BEM_DWACT(2);
    else if (mode == 1) BEM_DWACT(1);
    else BEM_DWACT(0);
#undef BEM_DWACT
    return bem_check_launch("dwact_bwd");
}

extern "C" int bem_grad_sumsq_f32(const float* g, int64_t n, double* acc, void* stream) {
    BEM_REQUIRE(g && acc && n > 0 && ((uintptr_t)g & 15) == 0, "grad_sumsq: bad arguments (16-byte aligned buffer)");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double), s) != hipSuccess) return bem_check_launch("grad_sumsq memset");
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n, 1024), 1024);
    sumsq_kernel<<<grid, 256, 0, s>>>(g, n, acc);
    return bem_check_launch("grad_sumsq");
}

extern "C" int bem_adamw_step_f32(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, int step, float max_norm, const double* sumsq, float* norm_out, const float* hyper, void* stream) {
    BEM_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || hyper), "adamw_step: bad arguments");
    BEM_REQUIRE(max_norm <= 0.f || sumsq, "adamw_step: clipping needs the sum of squares from bem_grad_sumsq_f32");
    const double bc1 = 1.0 - pow((double)beta1, step > 0 ? step : 1), bc2 = 1.0 - pow((double)beta2, step > 0 ? step : 1);
    adamw_kernel<<<GRID1D(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                                                          max_norm, sumsq, norm_out, hyper);
    return bem_check_launch("adamw_step");
}
