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
