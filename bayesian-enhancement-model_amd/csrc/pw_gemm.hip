// Pointwise (1x1) channel-mix GEMM on the CDNA4 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[b][m][p] = act(sum_k W[m][k] * pro(x)[b][k][p] + bias[m]) + res[b][m][p]
//
// Mapping (NCHW stays NCHW, pixels on lanes):
//   * one wave = 128 consecutive pixels x (MTW * 32) output channels.  The 128 pixels are 4 interleaved
//     MFMA N-tiles: lane l owns pixels p0 + 4*(l & 31) + v, v = 0..3, so one float4 load per lane per
//     k-step feeds the B operand of all four tiles and the epilogue writes float4s.
//   * k-step s covers input channels 2s and 2s+1; lanes 0-31 hold channel 2s, lanes 32-63 channel 2s+1
//     (the B[k = l>>5][j = l&31] layout of the 32x32x2 instruction).
//   * A operand (weights) comes pre-packed as Wp[mtile][kstep][lane] = W[32*mtile + (lane&31)][2*kstep + (lane>>5)]
//     so a k-step is one coalesced 256-byte load per M-tile, L1/L2 resident.
//   * D layout: column = lane & 31 (pixel), row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
// LayerNorm prologue: per-pixel statistics over the K channels are taken in a first sweep (two passes:
// mean, then centred variance); the two half-waves each see half of the channels, one lane^32 exchange
// completes them.  The GEMM sweep re-reads x (L1/L2 hits) and normalises on the fly.
#include "bem_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PwK {
    const float* x1; const float* x2; int C1; int C2; int in_mode;
    const float* ln_w; const float* ln_b; float ln_eps;
    const float* Wp; int64_t w_bstride;
    const float* bias; int64_t bias_bstride;
    const float* res; const float* prelu; int act;
    float* out; int out_mode; int Win;
    int M; int K; int L; int KS; int MT;
};

__device__ __forceinline__ float4 ld4(const float* __restrict__ row, int p, int L, bool vec) {
    if (vec) return *reinterpret_cast<const float4*>(row + p);   // caller guarantees p + 3 < L
    float4 r;
    r.x = p < L ? row[p] : 0.f;
    r.y = p + 1 < L ? row[p + 1] : 0.f;
    r.z = p + 2 < L ? row[p + 2] : 0.f;
    r.w = p + 3 < L ? row[p + 3] : 0.f;
    return r;
}

// value of pro-input channel ch at this lane's 4 pixels (before LayerNorm)
__device__ __forceinline__ float4 load_x(const PwK& k, int b, int ch, int p, bool vec) {
    if (ch >= k.K) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (k.in_mode == 2 && ch >= k.C1) {
        return ld4(k.x2 + ((int64_t)b * k.C2 + (ch - k.C1)) * k.L, p, k.L, vec);
    }
    float4 v = ld4(k.x1 + ((int64_t)b * k.C1 + ch) * k.L, p, k.L, vec);
    if (k.in_mode == 1) {
        const float4 w = ld4(k.x2 + ((int64_t)b * k.C2 + ch) * k.L, p, k.L, vec);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    return v;
}

template <int MTW>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(PwK k) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int b = blockIdx.z;
    const int mt0 = blockIdx.y * MTW;
    const int p = (blockIdx.x * 4 + wave) * 128 + 4 * (lane & 31);
    const int half = lane >> 5;
    if (p - 4 * (lane & 31) >= k.L) return;   // whole wave beyond the plane (wave-uniform)
    const bool vec = (k.L % 4 == 0) && (p + 3 < k.L);
    const bool any = p < k.L;

    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f), rstd = make_float4(1.f, 1.f, 1.f, 1.f);
    const bool ln = k.ln_w != nullptr;
    if (ln) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int st = 0; st < k.KS; ++st) {
            const float4 v = any ? load_x(k, b, 2 * st + half, p, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64);
        s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
        const float inv = 1.f / (float)k.K;
        mean = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int st = 0; st < k.KS; ++st) {
            const int ch = 2 * st + half;
            if (ch < k.K) {
                const float4 v = any ? load_x(k, b, ch, p, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float dx = v.x - mean.x, dy = v.y - mean.y, dz = v.z - mean.z, dw = v.w - mean.w;
                q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
            }
        }
        q.x += __shfl_xor(q.x, 32, 64); q.y += __shfl_xor(q.y, 32, 64);
        q.z += __shfl_xor(q.z, 32, 64); q.w += __shfl_xor(q.w, 32, 64);
        rstd = make_float4(1.f / sqrtf(q.x * inv + k.ln_eps), 1.f / sqrtf(q.y * inv + k.ln_eps),
                           1.f / sqrtf(q.z * inv + k.ln_eps), 1.f / sqrtf(q.w * inv + k.ln_eps));
    }

    f32x16 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][v][r] = 0.f;

    const float* wp = k.Wp + (int64_t)b * k.w_bstride + ((int64_t)mt0 * k.KS) * 64 + lane;
    const int64_t mt_stride = (int64_t)k.KS * 64;
    for (int st = 0; st < k.KS; ++st) {
        const int ch = 2 * st + half;
        float4 xv = any ? load_x(k, b, ch, p, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ln && ch < k.K) {
            const float g = k.ln_w[ch], be = k.ln_b[ch];
            xv.x = (xv.x - mean.x) * rstd.x * g + be;
            xv.y = (xv.y - mean.y) * rstd.y * g + be;
            xv.z = (xv.z - mean.z) * rstd.z * g + be;
            xv.w = (xv.w - mean.w) * rstd.w * g + be;
        }
        float av[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) av[m] = (mt0 + m < k.MT) ? wp[m * mt_stride + (int64_t)st * 64] : 0.f;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], xv.x, acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], xv.y, acc[m][1], 0, 0, 0);
            acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], xv.z, acc[m][2], 0, 0, 0);
            acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], xv.w, acc[m][3], 0, 0, 0);
        }
    }

    if (!any) return;
    const float slope = (k.act == 1) ? k.prelu[0] : 0.f;
    const float* bias = k.bias ? k.bias + (int64_t)b * k.bias_bstride : nullptr;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= k.M) continue;
            float o[4] = {acc[m][0][r], acc[m][1][r], acc[m][2][r], acc[m][3][r]};
            if (bias) {
                const float bv = bias[row];
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] += bv;
            }
            if (k.act == 1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v] = o[v] >= 0.f ? o[v] : slope * o[v];
            }
            if (k.out_mode == 0) {
                const int64_t base = ((int64_t)b * k.M + row) * k.L;
                if (k.res) {
                    const float4 rv = ld4(k.res + base, p, k.L, vec);
                    o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w;
                }
                if (vec) {
                    *reinterpret_cast<float4*>(k.out + base + p) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (p + v < k.L) k.out[base + p + v] = o[v];
                }
            } else {
                // ConvTranspose2d(k=2, s=2) scatter: row = (dy*2+dx)*Co + co
                const int Co = k.M >> 2;
                const int q = row / Co, co = row - q * Co;
                const int dy = q >> 1, dx = q & 1;
                const int Hin = k.L / k.Win;
                const int64_t obase = ((int64_t)b * Co + co) * (4 * (int64_t)k.L);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int pp = p + v;
                    if (pp < k.L) {
                        const int yy = pp / k.Win, xx = pp - yy * k.Win;
                        k.out[obase + (int64_t)(2 * yy + dy) * (2 * k.Win) + (2 * xx + dx)] = o[v];
                    }
                }
                (void)Hin;
            }
        }
    }
}

__global__ void pack_pw_weight_kernel(const float* __restrict__ W, float* __restrict__ Wp, int M, int K, int MT, int KS) {
    // grid: (ceil(MT*KS*64 / 256), nsets)
    const int64_t per = (int64_t)MT * KS * 64;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    const int lane = (int)(i & 63);
    const int64_t t = i >> 6;
    const int st = (int)(t % KS), mt = (int)(t / KS);
    const int row = mt * 32 + (lane & 31), col = 2 * st + (lane >> 5);
    const int set = blockIdx.y;
    Wp[(int64_t)set * per + i] = (row < M && col < K) ? W[((int64_t)set * M + row) * K + col] : 0.f;
}

}  // namespace

extern "C" int64_t bem_pw_packed_elems(int M, int K) { return (int64_t)cdiv(M, 32) * cdiv(K, 2) * 64; }

extern "C" int bem_pack_pw_weight_f32(const float* W, float* Wp, int nsets, int M, int K, void* stream) {
    BEM_REQUIRE(W && Wp, "pack_pw_weight: null tensor");
    BEM_REQUIRE(nsets >= 0 && nsets <= 65535 && M > 0 && K > 0, "pack_pw_weight: bad shape");
    if (nsets == 0) return BEM_OK;
    const int MT = cdiv(M, 32), KS = cdiv(K, 2);
    dim3 grid((unsigned)cdiv64((int64_t)MT * KS * 64, 256), nsets);
    pack_pw_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(W, Wp, M, K, MT, KS);
    return bem_check_launch("pack_pw_weight");
}

extern "C" int bem_pw_gemm_f32(const bem_pw_args* a, void* stream) {
    BEM_REQUIRE(a, "pw_gemm: null args");
    BEM_REQUIRE(a->x1 && a->Wp && a->out, "pw_gemm: null tensor");
    BEM_REQUIRE(a->B >= 0 && a->B <= 65535 && a->M > 0 && a->K > 0 && a->L >= 0, "pw_gemm: bad shape B=%d M=%d K=%d L=%d", a->B, a->M, a->K, a->L);
    BEM_REQUIRE(a->in_mode >= 0 && a->in_mode <= 2, "pw_gemm: in_mode %d", a->in_mode);
    if (a->in_mode == 0) BEM_REQUIRE(a->K == a->C1, "pw_gemm: K %d != C1 %d", a->K, a->C1);
    if (a->in_mode == 1) BEM_REQUIRE(a->x2 && a->K == a->C1 && a->C1 == a->C2, "pw_gemm: sum mode needs x2 and K == C1 == C2");
    if (a->in_mode == 2) BEM_REQUIRE(a->x2 && a->K == a->C1 + a->C2, "pw_gemm: cat mode needs x2 and K == C1 + C2");
    BEM_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr), "pw_gemm: ln_w / ln_b must both be set or both NULL");
    BEM_REQUIRE(a->act == 0 || (a->act == 1 && a->prelu), "pw_gemm: act %d", a->act);
    BEM_REQUIRE(a->out_mode == 0 || (a->out_mode == 1 && a->M % 4 == 0 && a->Win > 0 && a->L % a->Win == 0 && !a->res),
                "pw_gemm: out_mode %d constraints", a->out_mode);
    if (a->B == 0 || a->L == 0) return BEM_OK;
    PwK k;
    k.x1 = a->x1; k.x2 = a->x2; k.C1 = a->C1; k.C2 = a->C2; k.in_mode = a->in_mode;
    k.ln_w = a->ln_w; k.ln_b = a->ln_b; k.ln_eps = a->ln_eps;
    k.Wp = a->Wp; k.w_bstride = a->w_bstride; k.bias = a->bias; k.bias_bstride = a->bias_bstride;
    k.res = a->res; k.prelu = a->prelu; k.act = a->act; k.out = a->out; k.out_mode = a->out_mode; k.Win = a->Win;
    k.M = a->M; k.K = a->K; k.L = a->L; k.KS = cdiv(a->K, 2); k.MT = cdiv(a->M, 32);
    hipStream_t s = (hipStream_t)stream;
    if (k.MT == 1) {
        dim3 grid(cdiv(a->L, 512), 1, a->B);
        pw_gemm_kernel<1><<<grid, 256, 0, s>>>(k);
    } else {
        dim3 grid(cdiv(a->L, 512), cdiv(k.MT, 2), a->B);
        pw_gemm_kernel<2><<<grid, 256, 0, s>>>(k);
    }
    return bem_check_launch("pw_gemm");
}
