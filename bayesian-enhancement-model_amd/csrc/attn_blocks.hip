// The three small blocks DecompDualBranch adds around its bottlenecks (basicsr/archs/DecompModel_arch.py):
//   CrossFusionBlock  :57-66   x_tgt + gate * (W x_src + b)        -> the gate is folded into W and b here (row scaling, once per weight
//                                                                     version); the block itself is then one x6 GEMM with a residual
//   SEBlock           :68-83   x * sigmoid(W2 relu(W1 mean_hw(x)))  -> plane means (bem_plane_mean_f32) + se_gate_kernel = the (B,C) gate
//   SpatialAttention  :85-99   x * sigmoid(conv7x7([mean_c x, max_c x]))
// SE and attention follow each other on the same tensor (:318-324), so the attention kernels take the SE gate as a per-channel factor:
// the scaled tensor x * y is never written.  All of it is HBM-bound elementwise / reduction work on the deepest level's planes (H/4 x W/4):
// coalesced rows of one plane per wavefront access, channels as the loop.
#include "bem_common.h"

namespace {

__global__ void row_scale_kernel(const float* __restrict__ w, const float* __restrict__ s, float* __restrict__ out, int64_t total, int K) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) out[i] = w[i] * s[i / K];
}

// one workgroup per image: hidden = relu(W1 mean) (a wavefront per hidden unit), y = sigmoid(W2 hidden) (a thread per channel)
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ mean, const float* __restrict__ w1, const float* __restrict__ w2,
                                                      float* __restrict__ y, int C, int Cr) {
    extern __shared__ float sm[];              // [C] means | [Cr] hidden
    float* m = sm;
    float* h = sm + C;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += 256) m[c] = mean[(int64_t)b * C + c];
    __syncthreads();
    for (int r = wave; r < Cr; r += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(w1[(int64_t)r * C + c], m[c], acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, BEM_WAVE);
        if (lane == 0) h[r] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.f;
        for (int r = 0; r < Cr; ++r) acc = fmaf(w2[(int64_t)c * Cr + r], h[r], acc);
        y[(int64_t)b * C + c] = 1.f / (1.f + expf(-acc));
    }
}

// a thread per pixel: mean and max over the channels of x * y  ->  map (B, 2, HW)
__global__ void sa_stats_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ map, int C, int64_t HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / HW, p = i - b * HW;
    const float* xp = x + b * C * HW + p;
    const float* yb = y ? y + b * C : nullptr;
    float s = 0.f, mx = -INFINITY;
    for (int c = 0; c < C; ++c) {
        const float v = xp[(int64_t)c * HW] * (yb ? yb[c] : 1.f);
        s += v;
        mx = fmaxf(mx, v);
    }
    map[(b * 2) * HW + p] = s / (float)C;
    map[(b * 2 + 1) * HW + p] = mx;
}

// a thread per pixel: a = sigmoid(conv_kxk(map), zero padding k / 2); out[c] = x[c] * y[c] * a
__global__ void sa_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ map, const float* __restrict__ w,
                                float* __restrict__ out, int C, int H, int W, int k, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t HW = (int64_t)H * W, b = i / HW, p = i - b * HW;
    const int py = (int)(p / W), px = (int)(p - (int64_t)py * W), r = k / 2;
    float acc = 0.f;
    for (int ch = 0; ch < 2; ++ch) {
        const float* mp = map + (b * 2 + ch) * HW;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = py + dy - r;
            if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = px + dx - r;
                if (xx >= 0 && xx < W) acc = fmaf(w[(ch * k + dy) * k + dx], mp[(int64_t)yy * W + xx], acc);
            }
        }
    }
    const float a = 1.f / (1.f + expf(-acc));
    const float* xp = x + b * C * HW + p;
    float* op = out + b * C * HW + p;
    const float* yb = y ? y + b * C : nullptr;
    for (int c = 0; c < C; ++c) op[(int64_t)c * HW] = xp[(int64_t)c * HW] * (yb ? yb[c] : 1.f) * a;
}

// ------------------------------------------------------------------------------------------------------------------------
// Training side of the three blocks (the planes are those of the deepest level: every kernel here is a few microseconds of work).
//   out = (add ? add : 0) + scale[b or 0][c] * x  (+ add_bc[b][c] * add_bc_scale)        chan_scale_kernel
//   out[b][c] = sum_p a b   (per image)   /   out[c] += sum_{b,p} a b   (a parameter's gradient)      chan_dot_kernel
// ------------------------------------------------------------------------------------------------------------------------
__global__ void chan_scale_kernel(const float* __restrict__ x, const float* __restrict__ scale, int64_t scale_bs, const float* __restrict__ add,
                                  const float* __restrict__ add_bc, float add_bc_scale, float* __restrict__ out, int C, int64_t HW, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t bc = i / HW, b = bc / C;
    const int c = (int)(bc - b * C);
    float v = scale[b * scale_bs + c] * x[i];
    if (add) v += add[i];
    if (add_bc) v = fmaf(add_bc[bc], add_bc_scale, v);
    out[i] = v;
}

// grid (C, per_batch ? B : 1): one workgroup per (image,) channel
__global__ __launch_bounds__(256) void chan_dot_kernel(const float* __restrict__ a, const float* __restrict__ b2, float* __restrict__ out, int Bn, int C,
                                                       int64_t HW, int per_batch) {
    __shared__ float sh[4];
    const int c = blockIdx.x;
    float acc = 0.f;
    const int b_lo = per_batch ? blockIdx.y : 0, b_hi = per_batch ? blockIdx.y + 1 : Bn;
    for (int b = b_lo; b < b_hi; ++b) {
        const float* ap = a + ((int64_t)b * C + c) * HW;
        const float* bp = b2 + ((int64_t)b * C + c) * HW;
        for (int64_t p = threadIdx.x; p < HW; p += 256) acc = fmaf(ap[p], bp[p], acc);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, BEM_WAVE);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = sh[0] + sh[1] + sh[2] + sh[3];
        if (per_batch) out[(int64_t)blockIdx.y * C + c] = s;
        else atomicAdd(out + c, s);
    }
}

// SEBlock's gate backward, one workgroup per image: y = sigmoid(z2), z2 = W2 h, h = relu(W1 m):  dz2 = dy y (1 - y); dW2 += dz2 h^T; dh = W2^T dz2;
// dz1 = dh [h > 0]; dW1 += dz1 m^T; dm = W1^T dz1.
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ w1, const float* __restrict__ w2,
                                                          const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dmean,
                                                          float* __restrict__ dw1, float* __restrict__ dw2, int C, int Cr) {
    extern __shared__ float sm[];              // [C] m | [Cr] h | [C] dz2 | [Cr] dz1
    float* m = sm;
    float* h = m + C;
    float* dz2 = h + Cr;
    float* dz1 = dz2 + C;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < C; c += 256) {
        m[c] = mean[(int64_t)b * C + c];
        const float yv = y[(int64_t)b * C + c];
        dz2[c] = dy[(int64_t)b * C + c] * yv * (1.f - yv);
    }
    __syncthreads();
    for (int r = wave; r < Cr; r += 4) {
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(w1[(int64_t)r * C + c], m[c], acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, BEM_WAVE);
        if (lane == 0) h[r] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int r = wave; r < Cr; r += 4) {       // dh[r] = sum_c W2[c][r] dz2[c]
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) acc = fmaf(w2[(int64_t)c * Cr + r], dz2[c], acc);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, BEM_WAVE);
        if (lane == 0) dz1[r] = h[r] > 0.f ? acc : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float dm = 0.f;
        for (int r = 0; r < Cr; ++r) {
            atomicAdd(dw2 + (int64_t)c * Cr + r, dz2[c] * h[r]);
            atomicAdd(dw1 + (int64_t)r * C + c, dz1[r] * m[c]);
            dm = fmaf(w1[(int64_t)r * C + c], dz1[r], dm);
        }
        dmean[(int64_t)b * C + c] = dm;
    }
}

__device__ __forceinline__ float sa_conv(const float* __restrict__ map, const float* __restrict__ w, int64_t b, int py, int px, int H, int W, int k) {
    const int64_t HW = (int64_t)H * W;
    const int r = k / 2;
    float acc = 0.f;
    for (int ch = 0; ch < 2; ++ch) {
        const float* mp = map + (b * 2 + ch) * HW;
        for (int dy = 0; dy < k; ++dy) {
            const int yy = py + dy - r;
            if (yy < 0 || yy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int xx = px + dx - r;
                if (xx >= 0 && xx < W) acc = fmaf(w[(ch * k + dy) * k + dx], mp[(int64_t)yy * W + xx], acc);
            }
        }
    }
    return acc;
}

// dpre[b][p] = (sum_c dout x) a (1 - a),  a = sigmoid(conv(map))
__global__ void sa_bwd_pre_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ map, const float* __restrict__ w,
                                  float* __restrict__ dpre, int C, int H, int W, int k, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t HW = (int64_t)H * W, b = i / HW, p = i - b * HW;
    const float a = 1.f / (1.f + expf(-sa_conv(map, w, b, (int)(p / W), (int)(p % W), H, W, k)));
    const float* xp = x + b * C * HW + p;
    const float* dp = dout + b * C * HW + p;
    float da = 0.f;
    for (int c = 0; c < C; ++c) da = fmaf(dp[(int64_t)c * HW], xp[(int64_t)c * HW], da);
    dpre[i] = da * a * (1.f - a);
}

// dx[c] = dout[c] a + dmap_mean / C + [c == argmax_c x] dmap_max,  dmap[ch][p] = sum_taps w[ch][tap] dpre[p - tap offset]   (transposed conv)
__global__ void sa_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ map, const float* __restrict__ w,
                                 const float* __restrict__ dpre, float* __restrict__ dx, int C, int H, int W, int k, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t HW = (int64_t)H * W, b = i / HW, p = i - b * HW;
    const int py = (int)(p / W), px = (int)(p % W), r = k / 2;
    const float a = 1.f / (1.f + expf(-sa_conv(map, w, b, py, px, H, W, k)));
    float dmean = 0.f, dmax = 0.f;
    const float* dq = dpre + b * HW;
    for (int dy = 0; dy < k; ++dy) {
        const int yy = py - (dy - r);                      // the output pixel whose window holds this pixel at tap (dy, dx)
        if (yy < 0 || yy >= H) continue;
        for (int dxx = 0; dxx < k; ++dxx) {
            const int xx = px - (dxx - r);
            if (xx < 0 || xx >= W) continue;
            const float g = dq[(int64_t)yy * W + xx];
            dmean = fmaf(w[(0 * k + dy) * k + dxx], g, dmean);
            dmax = fmaf(w[(1 * k + dy) * k + dxx], g, dmax);
        }
    }
    const float* xp = x + b * C * HW + p;
    const float* dp = dout + b * C * HW + p;
    float* op = dx + b * C * HW + p;
    int am = 0;
    float mx = xp[0];
    for (int c = 1; c < C; ++c) {
        const float v = xp[(int64_t)c * HW];
        if (v > mx) { mx = v; am = c; }                     // first maximum, as torch.max(dim) reports it
    }
    const float dmc = dmean / (float)C;
    for (int c = 0; c < C; ++c) op[(int64_t)c * HW] = dp[(int64_t)c * HW] * a + dmc + (c == am ? dmax : 0.f);
}

// dw[ch][dy][dx] += sum_{b,p} map[b][ch][p + tap offset] dpre[b][p]        grid = 2 k k workgroups
__global__ __launch_bounds__(256) void sa_bwd_dw_kernel(const float* __restrict__ map, const float* __restrict__ dpre, float* __restrict__ dw, int Bn,
                                                        int H, int W, int k) {
    __shared__ float sh[4];
    const int tap = blockIdx.x, ch = tap / (k * k), dy = (tap / k) % k, dxx = tap % k, r = k / 2;
    const int64_t HW = (int64_t)H * W;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < (int64_t)Bn * HW; i += 256) {
        const int64_t b = i / HW, p = i - b * HW;
        const int yy = (int)(p / W) + dy - r, xx = (int)(p % W) + dxx - r;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc = fmaf(map[(b * 2 + ch) * HW + (int64_t)yy * W + xx], dpre[i], acc);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, BEM_WAVE);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dw + tap, sh[0] + sh[1] + sh[2] + sh[3]);
}

}  // namespace

extern "C" int bem_row_scale_f32(const float* w, const float* scale, float* out, int M, int K, void* stream) {
    BEM_REQUIRE(w && scale && out && M > 0 && K > 0, "row_scale: bad arguments");
    const int64_t total = (int64_t)M * K;
    row_scale_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, (hipStream_t)stream>>>(w, scale, out, total, K);
    return bem_check_launch("row_scale");
}

extern "C" int bem_se_gate_f32(const float* mean, const float* w1, const float* w2, float* y, int B, int C, int Cr, void* stream) {
    BEM_REQUIRE(mean && w1 && w2 && y, "se_gate: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && Cr > 0 && (size_t)(C + Cr) * sizeof(float) <= 48 * 1024, "se_gate: bad shape (B=%d, C=%d, C/r=%d)", B, C, Cr);
    if (B == 0) return BEM_OK;
    se_gate_kernel<<<B, 256, (size_t)(C + Cr) * sizeof(float), (hipStream_t)stream>>>(mean, w1, w2, y, C, Cr);
    return bem_check_launch("se_gate");
}

extern "C" int bem_spatial_attention_f32(const float* x, const float* chan_scale, const float* w, float* map_ws, float* out, int B, int C, int H,
                                         int W, int k, void* stream) {
    BEM_REQUIRE(x && w && map_ws && out, "spatial_attention: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && (k == 3 || k == 7), "spatial_attention: bad shape / kernel size %d (3 or 7)", k);
    if (B == 0) return BEM_OK;
    const int64_t total = (int64_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    sa_stats_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, s>>>(x, chan_scale, map_ws, C, (int64_t)H * W, total);
    sa_apply_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, s>>>(x, chan_scale, map_ws, w, out, C, H, W, k, total);
    return bem_check_launch("spatial_attention");
}

// out = scale[b * scale_bstride + c] * x (+ add) (+ add_bc[b][c] * add_bc_scale): scale_bstride = C for per-image factors (the SE gate), 0 for a
// parameter (CrossFusionBlock's gate).  add, add_bc: NULL or (B,C,HW) / (B,C).
extern "C" int bem_chan_scale_f32(const float* x, const float* scale, int64_t scale_bstride, const float* add, const float* add_bc, float add_bc_scale,
                                  float* out, int B, int C, int64_t HW, void* stream) {
    BEM_REQUIRE(x && scale && out && B >= 0 && C > 0 && HW > 0 && (scale_bstride == 0 || scale_bstride == C), "chan_scale: bad arguments");
    const int64_t total = (int64_t)B * C * HW;
    if (total == 0) return BEM_OK;
    chan_scale_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, (hipStream_t)stream>>>(x, scale, scale_bstride, add, add_bc, add_bc_scale, out, C, HW, total);
    return bem_check_launch("chan_scale");
}

// per_batch: out (B,C) = sum_p a b, written; otherwise out (C) += sum_{b,p} a b (accumulated: a parameter's gradient buffer)
extern "C" int bem_chan_dot_f32(const float* a, const float* b, float* out, int B, int C, int64_t HW, int per_batch, void* stream) {
    BEM_REQUIRE(a && b && out && B > 0 && C > 0 && HW > 0 && C <= 65535 && B <= 65535, "chan_dot: bad arguments");
    chan_dot_kernel<<<dim3(C, per_batch ? B : 1), 256, 0, (hipStream_t)stream>>>(a, b, out, B, C, HW, per_batch);
    return bem_check_launch("chan_dot");
}

extern "C" int bem_se_gate_bwd_f32(const float* mean, const float* w1, const float* w2, const float* y, const float* dy, float* dmean, float* dw1,
                                   float* dw2, int B, int C, int Cr, void* stream) {
    BEM_REQUIRE(mean && w1 && w2 && y && dy && dmean && dw1 && dw2, "se_gate_bwd: null tensor");
    BEM_REQUIRE(B > 0 && C > 0 && Cr > 0 && (size_t)(2 * C + 2 * Cr) * sizeof(float) <= 48 * 1024, "se_gate_bwd: bad shape");
    se_gate_bwd_kernel<<<B, 256, (size_t)(2 * C + 2 * Cr) * sizeof(float), (hipStream_t)stream>>>(mean, w1, w2, y, dy, dmean, dw1, dw2, C, Cr);
    return bem_check_launch("se_gate_bwd");
}

// backward of bem_spatial_attention_f32 with chan_scale = NULL: x, dout, dx (B,C,H,W); map: the (B,2,H,W) workspace the forward filled;
// dpre_ws: (B,H,W) floats; dw (1,2,k,k) accumulated.
extern "C" int bem_spatial_attention_bwd_f32(const float* x, const float* dout, const float* map, const float* w, float* dpre_ws, float* dx, float* dw,
                                             int B, int C, int H, int W, int k, void* stream) {
    BEM_REQUIRE(x && dout && map && w && dpre_ws && dx && dw, "spatial_attention_bwd: null tensor");
    BEM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (k == 3 || k == 7), "spatial_attention_bwd: bad shape");
    const int64_t total = (int64_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    sa_bwd_pre_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, s>>>(x, dout, map, w, dpre_ws, C, H, W, k, total);
    sa_bwd_dx_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, s>>>(x, dout, map, w, dpre_ws, dx, C, H, W, k, total);
    sa_bwd_dw_kernel<<<2 * k * k, 256, 0, s>>>(map, dpre_ws, dw, B, H, W, k);
    return bem_check_launch("spatial_attention_bwd");
}
