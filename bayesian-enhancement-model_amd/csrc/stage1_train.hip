// Stage-I (Bayesian condition generator) training: the pieces of ConditionGenerator.optimize_parameters
// (basicsr/models/condition_generator_model.py:176-218) that the Stage-II training kernels do not already cover.
//   * Bayesian leaves (basicsr/bayesian/conv.py:84-114, linear.py:61-90, base_layer.py:26-40): threshold-EMA prior update, KL(q || prior)
//     value and gradient, and the reparameterisation gradient  w = mu + softplus(rho) eps  ->  dmu += dw, drho += dw eps sigmoid(rho);
//   * masked-image-modelling token mix (basicsr/archs/UNet_arch.py:463-466) forward / backward;
//   * backward of the Stage-I U-Net's resampling layers: PatchMerging gather (UNet_arch.py:74-78), bilinear x s up-sampling and PReLU
//     (DualUpSample, UNet_arch.py:97-127).
// Stage-I planes are H/16 x W/16 (8 x 8 at the shipped gt_size 128): every kernel here is launch-latency sized, so they are plain
// one-thread-per-element kernels with block reductions + one atomic per block where a sum is needed.
#include "bem_common.h"
#include <cstring>
#define BEM_STEP_WORDS_MAX 512

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ float softplus_ref(float r) { return log1pf(expf(r)); }     // torch.log1p(torch.exp(rho)), conv.py:101
__device__ __forceinline__ float sigmoid_ref(float r) { return 1.f / (1.f + expf(-r)); }

__global__ void prior_ema_kernel(float* __restrict__ pmu, float* __restrict__ prho, const float* __restrict__ mu,
                                 const float* __restrict__ rho, float decay, const float* __restrict__ decay_dev, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    if (decay_dev) decay = decay_dev[0];                 // the iteration's decay read from HBM (captured step)
    pmu[i] = decay * pmu[i] + (1.f - decay) * mu[i];
    prho[i] = decay * prho[i] + (1.f - decay) * rho[i];
}

// kl.mean() of  log(sp) - log(sq) + (sq^2 + (mq - mp)^2) / (2 sp^2) - 0.5   (base_layer.py:38-39), added to out[0]
__global__ void kl_kernel(const float* __restrict__ mu, const float* __restrict__ rho, const float* __restrict__ pmu,
                          const float* __restrict__ prho, int64_t n, float inv_n, float* __restrict__ out) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float sq = softplus_ref(rho[i]), sp = softplus_ref(prho[i]), d = mu[i] - pmu[i];
        acc += logf(sp) - logf(sq) + (sq * sq + d * d) / (2.f * sp * sp) - 0.5f;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, s * inv_n);
}

// d(scale * kl.mean()) : dmu += s (mq - mp) / sp^2,  drho += s (sq / sp^2 - 1 / sq) sigmoid(rho),  s = g[0] / n
__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ rho, const float* __restrict__ pmu,
                              const float* __restrict__ prho, int64_t n, float inv_n, const float* __restrict__ g,
                              float* __restrict__ dmu, float* __restrict__ drho) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const float s = g[0] * inv_n;
    const float sq = softplus_ref(rho[i]), sp = softplus_ref(prho[i]);
    const float isp2 = 1.f / (sp * sp);
    dmu[i] += s * (mu[i] - pmu[i]) * isp2;
    drho[i] += s * (sq * isp2 - 1.f / sq) * sigmoid_ref(rho[i]);
}

__global__ void reparam_bwd_kernel(const float* __restrict__ gw, const float* __restrict__ eps, const float* __restrict__ rho,
                                   float* __restrict__ dmu, float* __restrict__ drho, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const float g = gw[i];
    dmu[i] += g;
    drho[i] += g * eps[i] * sigmoid_ref(rho[i]);
}

// ------------------------------------------------------------------------------------------------------------------------
// The same four steps for ALL Bayesian tensors of a net in one launch each (60 leaves / 90 tensors in the shipped Stage-I net: per tensor
// they are ~630 launches of a few hundred elements per training step).  A "bank" describes the tensors as segments of flat arenas:
//   seg[s] = { mu, rho, dmu, drho (device pointers of the parameter / its gradient buffer), off (first element in the arenas, a
//              multiple of 4), n (elements), stream counter (low bits of the tensor's Philox stream id) }
//   blk[b] = { segment, first element of the block inside the segment }         one workgroup = 1024 consecutive elements of one segment
//   arenas: prior_mu, prior_rho, w (the sample), eps (its draw), gw (the sample's gradient, zeroed by the sampling launch)
// Arithmetic per element is that of prior_ema_kernel / randn_kernel / bnn_sample_kernel / kl_kernel / kl_bwd_kernel / reparam_bwd_kernel.
// ------------------------------------------------------------------------------------------------------------------------
struct bank_seg { float* mu; float* rho; float* dmu; float* drho; int64_t off; int64_t n; uint64_t counter; float inv_n; int pad; };
static_assert(sizeof(bank_seg) == 8 * 8, "bank_seg is eight 64-bit words (bem.modules.BayesBank builds it as an int64 table)");
struct bank_blk { int32_t seg; int32_t first; };
constexpr int BANK_EPT = 4;            // elements per thread

__global__ __launch_bounds__(NT) void bank_sample_kernel(const bank_seg* __restrict__ segs, const bank_blk* __restrict__ blks,
                                                         float* __restrict__ pmu, float* __restrict__ prho, float* __restrict__ w,
                                                         float* __restrict__ eps, float* __restrict__ gw, float decay,
                                                         const float* __restrict__ decay_dev, uint64_t seed, uint64_t stream_base,
                                                         const uint64_t* __restrict__ stream_add) {
    const bank_blk bk = blks[blockIdx.x];
    const bank_seg sg = segs[bk.seg];
    if (decay_dev) decay = decay_dev[0];
    uint64_t sid = stream_base + sg.counter;
    if (stream_add) sid += stream_add[0];
#pragma unroll
    for (int q = 0; q < BANK_EPT; ++q) {
        const int64_t i = (int64_t)bk.first + q * NT + threadIdx.x;
        if (i >= sg.n) break;
        const int64_t a = sg.off + i;
        const float m = sg.mu[i], r = sg.rho[i];
        pmu[a] = decay * pmu[a] + (1.f - decay) * m;
        prho[a] = decay * prho[a] + (1.f - decay) * r;
        const float e = philox_normal(i, seed, sid);
        eps[a] = e;
        w[a] = m + log1pf(expf(r)) * e;
        gw[a] = 0.f;
    }
}

__global__ __launch_bounds__(NT) void bank_kl_kernel(const bank_seg* __restrict__ segs, const bank_blk* __restrict__ blks,
                                                     const float* __restrict__ pmu, const float* __restrict__ prho, float* __restrict__ out) {
    __shared__ float sh[4];
    const bank_blk bk = blks[blockIdx.x];
    const bank_seg sg = segs[bk.seg];
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < BANK_EPT; ++q) {
        const int64_t i = (int64_t)bk.first + q * NT + threadIdx.x;
        if (i < sg.n) {
            const float sq = softplus_ref(sg.rho[i]), sp = softplus_ref(prho[sg.off + i]), d = sg.mu[i] - pmu[sg.off + i];
            acc += logf(sp) - logf(sq) + (sq * sq + d * d) / (2.f * sp * sp) - 0.5f;
        }
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(out, s * sg.inv_n);
}

__global__ __launch_bounds__(NT) void bank_kl_bwd_kernel(const bank_seg* __restrict__ segs, const bank_blk* __restrict__ blks,
                                                         const float* __restrict__ pmu, const float* __restrict__ prho,
                                                         const float* __restrict__ g) {
    const bank_blk bk = blks[blockIdx.x];
    const bank_seg sg = segs[bk.seg];
    const float s = g[0] * sg.inv_n;
#pragma unroll
    for (int q = 0; q < BANK_EPT; ++q) {
        const int64_t i = (int64_t)bk.first + q * NT + threadIdx.x;
        if (i >= sg.n) break;
        const float r = sg.rho[i];
        const float sq = softplus_ref(r), sp = softplus_ref(prho[sg.off + i]);
        const float isp2 = 1.f / (sp * sp);
        sg.dmu[i] += s * (sg.mu[i] - pmu[sg.off + i]) * isp2;
        sg.drho[i] += s * (sq * isp2 - 1.f / sq) * sigmoid_ref(r);
    }
}

__global__ __launch_bounds__(NT) void bank_reparam_bwd_kernel(const bank_seg* __restrict__ segs, const bank_blk* __restrict__ blks,
                                                              const float* __restrict__ gw, const float* __restrict__ eps) {
    const bank_blk bk = blks[blockIdx.x];
    const bank_seg sg = segs[bk.seg];
#pragma unroll
    for (int q = 0; q < BANK_EPT; ++q) {
        const int64_t i = (int64_t)bk.first + q * NT + threadIdx.x;
        if (i >= sg.n) break;
        const float g = gw[sg.off + i];
        sg.dmu[i] += g;
        sg.drho[i] += g * eps[sg.off + i] * sigmoid_ref(sg.rho[i]);
    }
}

// fea * (1 - w) + token * w,  w = mask (B,H,W) broadcast over channels (UNet_arch.py:463-466)
__global__ void mask_token_kernel(const float* __restrict__ fea, const float* __restrict__ mask, const float* __restrict__ token,
                                  float* __restrict__ out, int C, int L, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % L), c = (int)((i / L) % C);
    const int64_t b = i / ((int64_t)L * C);
    const float w = mask[b * L + p];
    out[i] = fea[i] * (1.f - w) + token[c] * w;
}

// dfea = dout (1 - w);  dtoken[c] += sum_{b,p} dout w      grid (blocks over B*L, C)
__global__ void mask_token_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ mask, float* __restrict__ dfea,
                                      float* __restrict__ dtoken, int C, int L, int64_t BL) {
    __shared__ float sh[4];
    const int c = blockIdx.y;
    float acc = 0.f;
    for (int64_t j = (int64_t)blockIdx.x * NT + threadIdx.x; j < BL; j += (int64_t)gridDim.x * NT) {
        const int64_t b = j / L;
        const int p = (int)(j - b * L);
        const int64_t i = (b * C + c) * L + p;
        const float w = mask[j], d = dout[i];
        dfea[i] = d * (1.f - w);
        acc += d * w;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(dtoken + c, s);
}

// inverse gather of space_to_depth_kernel (elementwise.hip): dx (B,C,H,W)[2y+dy][2x+dx] = dy4 (B,4C,H/2,W/2)[(dy + 2 dx) C + c][y][x]
__global__ void depth_to_space_kernel(const float* __restrict__ d4, float* __restrict__ dx, int C, int H, int W, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int xx = (int)(i % W), y = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % C);
    const int64_t b = i / ((int64_t)W * H * C);
    const int q = (y & 1) + 2 * (xx & 1);
    const int h2 = H >> 1, w2 = W >> 1;
    dx[i] = d4[((b * 4 * C + (int64_t)q * C + c) * h2 + (y >> 1)) * w2 + (xx >> 1)];
}

__global__ void prelu_kernel(const float* __restrict__ x, const float* __restrict__ slope, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    out[i] = v >= 0.f ? v : slope[0] * v;
}

// dx = dout (x >= 0 ? 1 : a);  da += sum dout x [x < 0]     (nn.PReLU(), one shared slope)
__global__ void prelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ slope, const float* __restrict__ dout,
                                 float* __restrict__ dx, float* __restrict__ dslope, int64_t n) {
    __shared__ float sh[4];
    float acc = 0.f;
    const float a = slope[0];
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
        const float v = x[i], d = dout[i];
        dx[i] = v >= 0.f ? d : a * d;
        acc += v >= 0.f ? 0.f : d * v;
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(dslope, s);
}

// adjoint of bilinear_up_kernel (elementwise.hip): every output pixel hands its gradient to the four input pixels it read
__global__ void bilinear_up_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int C, int H, int W, int s, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    const int Wo = W * s, Ho = H * s;
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const int64_t bc = i / ((int64_t)Wo * Ho);
    const float rs = 1.f / (float)s;
    float sy = ((float)yo + 0.5f) * rs - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)xo + 0.5f) * rs - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    float* p = dx + bc * H * W;
    const float d = dout[i];
    atomicAdd(p + (int64_t)y0 * W + x0, hy * hx * d);
    atomicAdd(p + (int64_t)y0 * W + x1, hy * lx * d);
    atomicAdd(p + (int64_t)y1 * W + x0, ly * hx * d);
    atomicAdd(p + (int64_t)y1 * W + x1, ly * lx * d);
}

inline int grid_for(int64_t n, int cap = 1024) { return (int)std::min<int64_t>((n + NT - 1) / NT, cap); }

}  // namespace

// The per-iteration scalars of a captured step (Philox epoch, learning rate, Adam bias corrections, EMA decays) travel as kernel
// ARGUMENTS of this one launch into their device-resident slots: stream-ordered like every other launch, no pinned staging buffer
// whose reuse would have to be fenced.
namespace {
struct step_words { uint32_t w[BEM_STEP_WORDS_MAX]; };
__global__ void store_words_kernel(uint32_t* __restrict__ dst, step_words v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = v.w[i];
}
}  // namespace

extern "C" int bem_store_words(void* dst, const void* host_words, int n, void* stream) {
    BEM_REQUIRE(dst && host_words && n > 0 && n <= BEM_STEP_WORDS_MAX, "store_words: 1..%d 32-bit words", BEM_STEP_WORDS_MAX);
    step_words v;
    memcpy(v.w, host_words, sizeof(uint32_t) * (size_t)n);
    store_words_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>((uint32_t*)dst, v, n);
    return bem_check_launch("store_words");
}

extern "C" int bem_bnn_prior_ema_f32(float* prior_mu, float* prior_rho, const float* mu, const float* rho, float decay,
                                     const float* decay_dev, int64_t n, void* stream) {
    BEM_REQUIRE(prior_mu && prior_rho && mu && rho, "bnn_prior_ema: null tensor");
    BEM_REQUIRE(n >= 0 && decay >= 0.f && decay <= 1.f, "bnn_prior_ema: bad size / decay");
    if (n == 0) return BEM_OK;
    prior_ema_kernel<<<(unsigned)((n + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(prior_mu, prior_rho, mu, rho, decay, decay_dev, n);
    return bem_check_launch("bnn_prior_ema");
}

// Bank forms (see bank_seg above): segs = nseg x 8 int64 words, blks = nblk x 2 int32 words, both in device memory.
extern "C" int bem_bnn_bank_sample_f32(const void* segs, const void* blks, int nblk, float* prior_mu, float* prior_rho, float* w, float* eps,
                                       float* gw, float decay, const float* decay_dev, uint64_t seed, uint64_t stream_base,
                                       const uint64_t* stream_add, void* stream) {
    BEM_REQUIRE(segs && blks && prior_mu && prior_rho && w && eps && gw && nblk > 0, "bnn_bank_sample: bad arguments");
    BEM_REQUIRE(decay_dev || (decay >= 0.f && decay <= 1.f), "bnn_bank_sample: decay outside [0, 1]");
    bank_sample_kernel<<<nblk, NT, 0, (hipStream_t)stream>>>((const bank_seg*)segs, (const bank_blk*)blks, prior_mu, prior_rho, w, eps, gw, decay,
                                                            decay_dev, seed, stream_base, stream_add);
    return bem_check_launch("bnn_bank_sample");
}

extern "C" int bem_bnn_bank_kl_f32(const void* segs, const void* blks, int nblk, const float* prior_mu, const float* prior_rho, float* out,
                                   void* stream) {
    BEM_REQUIRE(segs && blks && prior_mu && prior_rho && out && nblk > 0, "bnn_bank_kl: bad arguments");
    bank_kl_kernel<<<nblk, NT, 0, (hipStream_t)stream>>>((const bank_seg*)segs, (const bank_blk*)blks, prior_mu, prior_rho, out);
    return bem_check_launch("bnn_bank_kl");
}

extern "C" int bem_bnn_bank_kl_bwd_f32(const void* segs, const void* blks, int nblk, const float* prior_mu, const float* prior_rho, const float* g,
                                       void* stream) {
    BEM_REQUIRE(segs && blks && prior_mu && prior_rho && g && nblk > 0, "bnn_bank_kl_bwd: bad arguments");
    bank_kl_bwd_kernel<<<nblk, NT, 0, (hipStream_t)stream>>>((const bank_seg*)segs, (const bank_blk*)blks, prior_mu, prior_rho, g);
    return bem_check_launch("bnn_bank_kl_bwd");
}

extern "C" int bem_bnn_bank_reparam_bwd_f32(const void* segs, const void* blks, int nblk, const float* gw, const float* eps, void* stream) {
    BEM_REQUIRE(segs && blks && gw && eps && nblk > 0, "bnn_bank_reparam_bwd: bad arguments");
    bank_reparam_bwd_kernel<<<nblk, NT, 0, (hipStream_t)stream>>>((const bank_seg*)segs, (const bank_blk*)blks, gw, eps);
    return bem_check_launch("bnn_bank_reparam_bwd");
}

extern "C" int bem_bnn_kl_f32(const float* mu, const float* rho, const float* prior_mu, const float* prior_rho, int64_t n, float* out,
                              void* stream) {
    BEM_REQUIRE(mu && rho && prior_mu && prior_rho && out, "bnn_kl: null tensor");
    BEM_REQUIRE(n > 0, "bnn_kl: empty tensor (the mean of the reference is undefined)");
    kl_kernel<<<grid_for(n, 256), NT, 0, (hipStream_t)stream>>>(mu, rho, prior_mu, prior_rho, n, 1.f / (float)n, out);
    return bem_check_launch("bnn_kl");
}

extern "C" int bem_bnn_kl_bwd_f32(const float* mu, const float* rho, const float* prior_mu, const float* prior_rho, int64_t n,
                                  const float* g, float* dmu, float* drho, void* stream) {
    BEM_REQUIRE(mu && rho && prior_mu && prior_rho && g && dmu && drho, "bnn_kl_bwd: null tensor");
    BEM_REQUIRE(n > 0, "bnn_kl_bwd: empty tensor");
    kl_bwd_kernel<<<(unsigned)((n + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(mu, rho, prior_mu, prior_rho, n, 1.f / (float)n, g, dmu, drho);
    return bem_check_launch("bnn_kl_bwd");
}

extern "C" int bem_bnn_reparam_bwd_f32(const float* gw, const float* eps, const float* rho, float* dmu, float* drho, int64_t n,
                                       void* stream) {
    BEM_REQUIRE(gw && eps && rho && dmu && drho, "bnn_reparam_bwd: null tensor");
    BEM_REQUIRE(n >= 0, "bnn_reparam_bwd: bad size");
    if (n == 0) return BEM_OK;
    reparam_bwd_kernel<<<(unsigned)((n + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(gw, eps, rho, dmu, drho, n);
    return bem_check_launch("bnn_reparam_bwd");
}

extern "C" int bem_mask_token_f32(const float* fea, const float* mask, const float* token, float* out, int B, int C, int H, int W,
                                  void* stream) {
    BEM_REQUIRE(fea && mask && token && out, "mask_token: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0, "mask_token: bad shape");
    const int64_t total = (int64_t)B * C * H * W;
    if (total == 0) return BEM_OK;
    mask_token_kernel<<<(unsigned)((total + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(fea, mask, token, out, C, H * W, total);
    return bem_check_launch("mask_token");
}

extern "C" int bem_mask_token_bwd_f32(const float* dout, const float* mask, float* dfea, float* dtoken, int B, int C, int H, int W,
                                      void* stream) {
    BEM_REQUIRE(dout && mask && dfea && dtoken, "mask_token_bwd: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && C <= 65535 && H > 0 && W > 0, "mask_token_bwd: bad shape");
    const int64_t BL = (int64_t)B * H * W;
    if (BL == 0) return BEM_OK;
    mask_token_bwd_kernel<<<dim3(grid_for(BL, 64), C), NT, 0, (hipStream_t)stream>>>(dout, mask, dfea, dtoken, C, H * W, BL);
    return bem_check_launch("mask_token_bwd");
}

extern "C" int bem_depth_to_space_f32(const float* d4, float* dx, int B, int C, int H, int W, void* stream) {
    BEM_REQUIRE(d4 && dx, "depth_to_space: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "depth_to_space: H, W must be even");
    const int64_t total = (int64_t)B * C * H * W;
    if (total == 0) return BEM_OK;
    depth_to_space_kernel<<<(unsigned)((total + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(d4, dx, C, H, W, total);
    return bem_check_launch("depth_to_space");
}

extern "C" int bem_prelu_f32(const float* x, const float* slope, float* out, int64_t n, void* stream) {
    BEM_REQUIRE(x && slope && out, "prelu: null tensor");
    BEM_REQUIRE(n >= 0, "prelu: bad size");
    if (n == 0) return BEM_OK;
    prelu_kernel<<<(unsigned)((n + NT - 1) / NT), NT, 0, (hipStream_t)stream>>>(x, slope, out, n);
    return bem_check_launch("prelu");
}

extern "C" int bem_prelu_bwd_f32(const float* x, const float* slope, const float* dout, float* dx, float* dslope, int64_t n, void* stream) {
    BEM_REQUIRE(x && slope && dout && dx && dslope, "prelu_bwd: null tensor");
    BEM_REQUIRE(n >= 0, "prelu_bwd: bad size");
    if (n == 0) return BEM_OK;
    prelu_bwd_kernel<<<grid_for(n, 256), NT, 0, (hipStream_t)stream>>>(x, slope, dout, dx, dslope, n);
    return bem_check_launch("prelu_bwd");
}

extern "C" int bem_bilinear_up_bwd_f32(const float* dout, float* dx, int B, int C, int H, int W, int s, void* stream) {
    BEM_REQUIRE(dout && dx, "bilinear_up_bwd: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && s >= 1, "bilinear_up_bwd: bad shape");
    const int64_t total = (int64_t)B * C * H * W * s * s;
    if (total == 0) return BEM_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * C * H * W, st) != hipSuccess) return bem_check_launch("bilinear_up_bwd memset");
    bilinear_up_bwd_kernel<<<(unsigned)((total + NT - 1) / NT), NT, 0, st>>>(dout, dx, C, H, W, s, total);
    return bem_check_launch("bilinear_up_bwd");
}
