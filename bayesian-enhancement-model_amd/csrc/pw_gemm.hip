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
#include <stdlib.h>

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
    int dbg;   // timing experiments (BEM_PW_DBG): bit0 skip MFMAs, bit1 skip epilogue
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

// ---- branch-free input fetch -------------------------------------------------------------------------------
// load_x above nests data-dependent branches (concat / sum / tail handling); inside an unrolled loop hipcc then
// emits every load in its own basic block followed by s_waitcnt vmcnt(0), i.e. one full memory latency PER LOAD.
// The forms below always load from a clamped, valid address and select the value afterwards, so an unrolled
// loop of them is a straight run of loads with a single wait.  SUM (x1 + x2) is a compile-time variant; concat
// is a pointer select.  `pc` must be a valid pixel index of the plane for every lane (clamp before calling).
template <bool SUM>
__device__ __forceinline__ float4 ldx4(const PwK& k, int b, int ch, int pc, bool keep) {
    const int chc = min(ch, k.K - 1);
    const bool first = chc < k.C1;
    const float* r1 = k.x1 + ((int64_t)b * k.C1 + (first ? chc : 0)) * k.L;
    const float* r2 = k.x2 + ((int64_t)b * k.C2 + (first ? 0 : chc - k.C1)) * k.L;
    const float* base = first ? r1 : r2;
    float4 v = *reinterpret_cast<const float4*>(base + pc);
    if (SUM) {
        const float4 w = *reinterpret_cast<const float4*>(k.x2 + ((int64_t)b * k.C2 + chc) * k.L + pc);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const bool m = keep && ch < k.K;
    v.x = m ? v.x : 0.f; v.y = m ? v.y : 0.f; v.z = m ? v.z : 0.f; v.w = m ? v.w : 0.f;
    return v;
}
// scalar form for planes whose length is not a multiple of 4 (rows are then unaligned): 4 clamped scalar loads
template <bool SUM>
__device__ __forceinline__ float4 ldx1(const PwK& k, int b, int ch, int p, bool keep) {
    const int chc = min(ch, k.K - 1);
    const bool first = chc < k.C1;
    const float* r1 = k.x1 + ((int64_t)b * k.C1 + (first ? chc : 0)) * k.L;
    const float* r2 = k.x2 + ((int64_t)b * k.C2 + (first ? 0 : chc - k.C1)) * k.L;
    const float* base = first ? r1 : r2;
    const float* sec = k.x2 + ((int64_t)b * k.C2 + chc) * k.L;
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = min(p + i, k.L - 1);
        float v = base[pi];
        if (SUM) v += sec[pi];
        o[i] = (keep && ch < k.K && p + i < k.L) ? v : 0.f;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}
// fills xr[0..N) with the k-steps s0, s0+1, ... of this lane (channel 2s + half).  VEC / SUM are compile-time: a
// uniform runtime branch around the loads would still end in a block join where hipcc waits vmcnt(0).
template <int N, bool VEC, bool SUM>
__device__ __forceinline__ void load_steps(const PwK& k, int b, int s0, int half, int p, bool any, float4 (&xr)[N]) {
    const int pc = any ? p : 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (VEC) xr[i] = ldx4<SUM>(k, b, 2 * (s0 + i) + half, pc, any);
        else xr[i] = ldx1<SUM>(k, b, 2 * (s0 + i) + half, pc, any);
    }
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


// ------------------------------------------------------------------------------------------------
// v2: LDS-staged variant.  One workgroup = 128 consecutive pixels x ALL output channels:
//   phase 0  the K x 128 input tile (after the sum / concat prologue) is pulled into LDS with every
//            load in flight at once (K > KCH is streamed in chunks of KCH channels);
//   phase 1  LayerNorm in LDS (one thread per pixel for the statistics, all threads normalise);
//   phase 2  MFMA: B operands come from LDS, A operands (packed weights) from L1/L2 with a register
//            double buffer 4 k-steps deep.
//            M-split (MT >= 4): wave w owns M-tiles w, w+4, ..., each against the 4 pixel-interleaved
//            N-tiles of the workgroup (one ds_read_b128 feeds 4 MFMAs per M-tile);
//            N-split (MT < 4):  wave w owns the 32 contiguous pixels [32w, 32w+32) against all M-tiles.
// x is read from HBM exactly once and LayerNorm is evaluated once per pixel, whatever M is.
// ------------------------------------------------------------------------------------------------
constexpr int PT = 128;      // pixels per workgroup
constexpr int KCH = 128;     // channels per LDS chunk when K is streamed

__device__ __forceinline__ void pw2_stage(const PwK& k, float* xs, int b, int p0, int c0, int kc, bool vecL) {
    // xs[(ch - c0) * PT + pix] for ch in [c0, c0 + kc), pix in [0, PT).  Loads are issued in batches of
    // UB float4 per thread before anything is written to LDS, so a batch costs one memory latency.
    constexpr int UB = 8;
    const int total = kc * (PT / 4);
    const bool sum = k.in_mode == 1;
    for (int base = 0; base < total; base += 256 * UB) {
        float4 v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int cc = idx / (PT / 4), p4 = idx - cc * (PT / 4);
            const int p = p0 + 4 * p4;
            const bool keep = idx < total && p < k.L;
            const int ch = keep ? c0 + cc : 0, pc = keep ? p : 0;
            if (vecL) v[u] = sum ? ldx4<true>(k, b, ch, pc, keep) : ldx4<false>(k, b, ch, pc, keep);
            else v[u] = sum ? ldx1<true>(k, b, ch, pc, keep) : ldx1<false>(k, b, ch, pc, keep);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            if (idx < total) *reinterpret_cast<float4*>(xs + (size_t)idx * 4) = v[u];
        }
    }
}

template <bool MSPLIT>
__global__ __launch_bounds__(256, 2) void pw_gemm2_kernel(PwK k, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                           // [kchunk][PT]
    float* stat = smem + (size_t)kchunk * PT;   // mean[PT], rstd[PT]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * PT;
    const bool vecL = (k.L % 4 == 0);
    const int Kp = 2 * k.KS;                    // K rounded up to even
    const int nchunks = (Kp + kchunk - 1) / kchunk;
    const bool ln = k.ln_w != nullptr;          // host guarantees nchunks == 1 with LayerNorm

    constexpr int NI = 2;                        // M-tiles per wave per group (M-split) 
    constexpr int NACC = MSPLIT ? NI * 4 : 3;
    const int ngroups = MSPLIT ? (k.MT + 4 * NI - 1) / (4 * NI) : 1;
    const float slope = (k.act == 1) ? k.prelu[0] : 0.f;
    const float* bias = k.bias ? k.bias + (int64_t)b * k.bias_bstride : nullptr;
    const float* wbase = k.Wp + (int64_t)b * k.w_bstride + lane;
    const int64_t mt_stride = (int64_t)k.KS * 64;

    for (int g = 0; g < ngroups; ++g) {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        int mts[MSPLIT ? NI : 3];
        if (MSPLIT) {
#pragma unroll
            for (int i = 0; i < NI; ++i) mts[i] = g * 4 * NI + i * 4 + wave;
        } else {
            mts[0] = 0; mts[1] = 1; mts[2] = 2;
        }
        for (int c = 0; c < nchunks; ++c) {
            const int c0 = c * kchunk;
            const int kc = min(kchunk, Kp - c0);
            if (c > 0 || g == 0 || nchunks > 1) {
                if (!(g > 0 && nchunks == 1)) {
                    __syncthreads();            // previous consumers of xs are done
                    pw2_stage(k, xs, b, p0, c0, kc, vecL);
                    __syncthreads();
                    if (ln) {
                        if (threadIdx.x < PT) {
                            float s = 0.f;
                            for (int ch = 0; ch < k.K; ++ch) s += xs[ch * PT + threadIdx.x];
                            const float mean = s / (float)k.K;
                            float q = 0.f;
                            for (int ch = 0; ch < k.K; ++ch) {
                                const float d = xs[ch * PT + threadIdx.x] - mean;
                                q = fmaf(d, d, q);
                            }
                            stat[threadIdx.x] = mean;
                            stat[PT + threadIdx.x] = 1.f / sqrtf(q / (float)k.K + k.ln_eps);
                        }
                        __syncthreads();
                        for (int idx = threadIdx.x; idx < k.K * PT; idx += 256) {
                            const int ch = idx / PT, pp = idx - ch * PT;
                            xs[idx] = (xs[idx] - stat[pp]) * stat[PT + pp] * k.ln_w[ch] + k.ln_b[ch];
                        }
                        __syncthreads();
                    }
                }
            }
            // ---- MFMA over this chunk ----
            const int ks0 = c0 >> 1, nks = kc >> 1;
            if (MSPLIT) {
                const float* w0 = (mts[0] < k.MT) ? wbase + mts[0] * mt_stride + (int64_t)ks0 * 64 : nullptr;
                const float* w1 = (mts[1] < k.MT) ? wbase + mts[1] * mt_stride + (int64_t)ks0 * 64 : nullptr;
                if (w0) {
                    float a0[4], a1[4];
                    const float* w1s = w1 ? w1 : w0;              // valid address either way; masked by multiplication
                    const float m1 = w1 ? 1.f : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int uc = min(u, nks - 1);
                        const float mk = (u < nks) ? 1.f : 0.f;
                        a0[u] = w0[uc * 64] * mk;
                        a1[u] = w1s[uc * 64] * (mk * m1);
                    }
                    for (int s0 = 0; s0 < nks; s0 += 4) {
                        float n0[4], n1[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sn = s0 + 4 + u;
                            const int sc = min(sn, nks - 1);
                            const float mk = (sn < nks) ? 1.f : 0.f;
                            n0[u] = w0[sc * 64] * mk;
                            n1[u] = w1s[sc * 64] * (mk * m1);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {      // branch-free: steps past the chunk pair zero weights with a valid LDS row
                            const int krow = min(2 * (s0 + u) + half, kc - 1);
                            const float4 xv = *reinterpret_cast<const float4*>(xs + krow * PT + 4 * j);
                            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], xv.x, acc[0], 0, 0, 0);
                            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], xv.y, acc[1], 0, 0, 0);
                            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], xv.z, acc[2], 0, 0, 0);
                            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], xv.w, acc[3], 0, 0, 0);
                            if (w1) {
                                acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], xv.x, acc[4], 0, 0, 0);
                                acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], xv.y, acc[5], 0, 0, 0);
                                acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], xv.z, acc[6], 0, 0, 0);
                                acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], xv.w, acc[7], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { a0[u] = n0[u]; a1[u] = n1[u]; }
                    }
                }
            } else {
                const float* w0 = wbase + (int64_t)ks0 * 64;
                float an[3][4];
#pragma unroll
                for (int m = 0; m < 3; ++m)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        an[m][u] = w0[(m < k.MT ? m : 0) * mt_stride + (int64_t)min(u, nks - 1) * 64] * ((m < k.MT && u < nks) ? 1.f : 0.f);
                for (int s0 = 0; s0 < nks; s0 += 4) {
                    float ac[3][4];
#pragma unroll
                    for (int m = 0; m < 3; ++m)
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            ac[m][u] = an[m][u];
                            const int sn = s0 + 4 + u;
                            an[m][u] = w0[(m < k.MT ? m : 0) * mt_stride + (int64_t)min(sn, nks - 1) * 64] * ((m < k.MT && sn < nks) ? 1.f : 0.f);
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {      // branch-free (zero weights past the chunk / past MT)
                        const int krow = min(2 * (s0 + u) + half, kc - 1);
                        const float xv = xs[krow * PT + 32 * wave + j];
#pragma unroll
                        for (int m = 0; m < 3; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[m][u], xv, acc[m], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue of this group ----
        // Residual values are fetched for a whole M-tile BEFORE anything is stored: `res` and `out` may alias as
        // far as the compiler knows, so interleaving them serialises one memory latency per row.
        constexpr int NM = MSPLIT ? NI : 3;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int mt = mts[m];
            if (mt >= k.MT) continue;
            const int rbase = mt * 32 + 4 * half;
            if (MSPLIT) {
                const int p = p0 + 4 * j;
                if (p >= k.L) continue;
                const bool vec = vecL && (p + 3 < k.L);
#pragma unroll
                for (int rh = 0; rh < 16; rh += 8) {       // residual rows fetched 8 at a time (register budget)
                    float4 rv[8];
                    if (k.res && k.out_mode == 0) {
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = rh + r8;
                            const int row = rbase + (r & 3) + 8 * (r >> 2);
                            rv[r8] = (row < k.M) ? ld4(k.res + ((int64_t)b * k.M + row) * k.L, p, k.L, vec) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        const int r = rh + r8;
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        if (row >= k.M) continue;
                        const float bv = bias ? bias[row] : 0.f;
                        float o[4] = {acc[m * 4 + 0][r] + bv, acc[m * 4 + 1][r] + bv, acc[m * 4 + 2][r] + bv, acc[m * 4 + 3][r] + bv};
                        if (k.act == 1) {
#pragma unroll
                            for (int v = 0; v < 4; ++v) o[v] = o[v] >= 0.f ? o[v] : slope * o[v];
                        }
                        if (k.out_mode == 0) {
                            const int64_t base = ((int64_t)b * k.M + row) * k.L;
                            if (k.res) { o[0] += rv[r8].x; o[1] += rv[r8].y; o[2] += rv[r8].z; o[3] += rv[r8].w; }
                            if (vec) {
                                *reinterpret_cast<float4*>(k.out + base + p) = make_float4(o[0], o[1], o[2], o[3]);
                            } else {
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    if (p + v < k.L) k.out[base + p + v] = o[v];
                            }
                        } else {
                            const int Co = k.M >> 2;
                            const int q = row / Co, co = row - q * Co;
                            const int64_t obase = ((int64_t)b * Co + co) * (4 * (int64_t)k.L);
#pragma unroll
                            for (int v = 0; v < 4; ++v) {
                                const int pp = p + v;
                                if (pp < k.L) {
                                    const int yy = pp / k.Win, xx = pp - yy * k.Win;
                                    k.out[obase + (int64_t)(2 * yy + (q >> 1)) * (2 * k.Win) + (2 * xx + (q & 1))] = o[v];
                                }
                            }
                        }
                    }
                }
            } else {
                const int p = p0 + 32 * wave + j;
                if (p >= k.L) continue;
                float rv[16];
                if (k.res && k.out_mode == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        rv[r] = (row < k.M) ? k.res[((int64_t)b * k.M + row) * k.L + p] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row >= k.M) continue;
                    float o = acc[m][r] + (bias ? bias[row] : 0.f);
                    if (k.act == 1) o = o >= 0.f ? o : slope * o;
                    if (k.out_mode == 0) {
                        if (k.res) o += rv[r];
                        k.out[((int64_t)b * k.M + row) * k.L + p] = o;
                    } else {
                        const int Co = k.M >> 2;
                        const int q = row / Co, co = row - q * Co;
                        const int yy = p / k.Win, xx = p - yy * k.Win;
                        k.out[((int64_t)b * Co + co) * (4 * (int64_t)k.L) + (int64_t)(2 * yy + (q >> 1)) * (2 * k.Win) + (2 * xx + (q & 1))] = o;
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// v3: barrier-free variants (each wave is an independent 128-pixel worker, as in v1).
//   pw_gemm3_reg<KSM, MTW>  K <= 2*KSM: the wave's whole input tile lives in registers (KSM float4 per lane,
//       all loads issued back to back = one memory latency), LayerNorm is evaluated on those registers, and the
//       wave then walks over ALL M-tiles, MTW at a time.  x is read once, LN computed once.
//   pw_gemm3_stream<MTW>    any K, no LayerNorm: x is streamed in batches of 8 k-steps, the next batch being
//       requested before the MFMAs of the current one (explicit software pipeline); grid.y walks M slices.
// ------------------------------------------------------------------------------------------------
template <int MTW>
__device__ __forceinline__ void pw3_epilogue(const PwK& k, int b, int mt0, int p, bool vec, int half,
                                             const f32x16 (&acc)[MTW][4]) {
    const float slope = (k.act == 1) ? k.prelu[0] : 0.f;
    // bias through an always-valid pointer + multiplicative mask: per-row `bias ? bias[row] : 0` behind `row < M`
    // compiles to a branch with a dependent load and a full vmcnt(0) wait for every row
    const float* bias = k.bias ? k.bias + (int64_t)b * k.bias_bstride : k.Wp;
    const float bmask = k.bias ? 1.f : 0.f;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (mt0 + m >= k.MT) continue;
        const int rbase = (mt0 + m) * 32 + 4 * half;
#pragma unroll
        for (int rh = 0; rh < 16; rh += 4) {
            float4 rv[4];
            float bv4[4];
#pragma unroll
            for (int r8 = 0; r8 < 4; ++r8) {
                const int r = rh + r8;
                bv4[r8] = bias[min(rbase + (r & 3) + 8 * (r >> 2), k.M - 1)] * bmask;
            }
            if (k.res && k.out_mode == 0) {
                if (vec) {
#pragma unroll
                    for (int r8 = 0; r8 < 4; ++r8) {      // clamped row: unconditional loads, rows >= M are never stored
                        const int r = rh + r8;
                        const int row = min(rbase + (r & 3) + 8 * (r >> 2), k.M - 1);
                        rv[r8] = *reinterpret_cast<const float4*>(k.res + ((int64_t)b * k.M + row) * k.L + p);
                    }
                } else {
#pragma unroll
                    for (int r8 = 0; r8 < 4; ++r8) {
                        const int r = rh + r8;
                        const int row = min(rbase + (r & 3) + 8 * (r >> 2), k.M - 1);
                        rv[r8] = ld4(k.res + ((int64_t)b * k.M + row) * k.L, p, k.L, false);
                    }
                }
            }
#pragma unroll
            for (int r8 = 0; r8 < 4; ++r8) {
                const int r = rh + r8;
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= k.M) continue;
                const float bv = bv4[r8];
                float o[4] = {acc[m][0][r] + bv, acc[m][1][r] + bv, acc[m][2][r] + bv, acc[m][3][r] + bv};
                if (k.act == 1) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) o[v] = o[v] >= 0.f ? o[v] : slope * o[v];
                }
                if (k.out_mode == 0) {
                    const int64_t base = ((int64_t)b * k.M + row) * k.L;
                    if (k.res) { o[0] += rv[r8].x; o[1] += rv[r8].y; o[2] += rv[r8].z; o[3] += rv[r8].w; }
                    if (vec) {
                        *reinterpret_cast<float4*>(k.out + base + p) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (p + v < k.L) k.out[base + p + v] = o[v];
                    }
                } else {
                    const int Co = k.M >> 2;
                    const int q = row / Co, co = row - q * Co;
                    const int64_t obase = ((int64_t)b * Co + co) * (4 * (int64_t)k.L);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int pp = p + v;
                        if (pp < k.L) {
                            const int yy = pp / k.Win, xx = pp - yy * k.Win;
                            k.out[obase + (int64_t)(2 * yy + (q >> 1)) * (2 * k.Win) + (2 * xx + (q & 1))] = o[v];
                        }
                    }
                }
            }
        }
    }
}

template <int KSM, int MTW, bool VEC, bool SUM>
__global__ __launch_bounds__(256, (KSM > 20 ? 1 : 2)) void pw_gemm3_reg_kernel(PwK k) {
    // LayerNorm scale / shift through LDS: read from global where they are used, each value costs an L1/L2 round trip
    // in front of the dependent normalisation (20 serialized loads per wave in the ISA)
    __shared__ float s_ln[4 * KSM];
    for (int i = threadIdx.x; i < 2 * KSM; i += 256) {
        const bool on = k.ln_w && i < k.K;
        s_ln[i] = on ? k.ln_w[min(i, k.K - 1)] : 0.f;
        s_ln[2 * KSM + i] = on ? k.ln_b[min(i, k.K - 1)] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int b = blockIdx.z;
    const int p = (blockIdx.x * 4 + wave) * 128 + 4 * (lane & 31);
    const bool vec = (k.L % 4 == 0) && (p + 3 < k.L);
    const bool any = p < k.L;
    float4 xr[KSM];
    load_steps<KSM, VEC, SUM>(k, b, 0, half, p, any, xr);      // channels >= K come back as zeros
    __syncthreads();                                            // s_ln visible (the only barrier; before any early exit)
    if (p - 4 * (lane & 31) >= k.L) return;
    if (k.ln_w) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int st = 0; st < KSM; ++st) { s.x += xr[st].x; s.y += xr[st].y; s.z += xr[st].z; s.w += xr[st].w; }
        s.x += __shfl_xor(s.x, 32, 64); s.y += __shfl_xor(s.y, 32, 64);
        s.z += __shfl_xor(s.z, 32, 64); s.w += __shfl_xor(s.w, 32, 64);
        const float inv = 1.f / (float)k.K;
        const float4 mean = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int st = 0; st < KSM; ++st) {
            const float mk = (2 * st + half < k.K) ? 1.f : 0.f;                // padded channels do not count
            const float dx = (xr[st].x - mean.x) * mk, dy = (xr[st].y - mean.y) * mk, dz = (xr[st].z - mean.z) * mk, dw = (xr[st].w - mean.w) * mk;
            q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
        q.x += __shfl_xor(q.x, 32, 64); q.y += __shfl_xor(q.y, 32, 64);
        q.z += __shfl_xor(q.z, 32, 64); q.w += __shfl_xor(q.w, 32, 64);
        const float4 rstd = make_float4(1.f / sqrtf(q.x * inv + k.ln_eps), 1.f / sqrtf(q.y * inv + k.ln_eps),
                                        1.f / sqrtf(q.z * inv + k.ln_eps), 1.f / sqrtf(q.w * inv + k.ln_eps));
#pragma unroll
        for (int st = 0; st < KSM; ++st) {
            const float g = s_ln[2 * st + half], be = s_ln[2 * KSM + 2 * st + half];      // 0 on padded channels
            xr[st].x = (xr[st].x - mean.x) * rstd.x * g + be;
            xr[st].y = (xr[st].y - mean.y) * rstd.y * g + be;
            xr[st].z = (xr[st].z - mean.z) * rstd.z * g + be;
            xr[st].w = (xr[st].w - mean.w) * rstd.w * g + be;
        }
    }
    const float* wbase = k.Wp + (int64_t)b * k.w_bstride + lane;
    const int64_t mt_stride = (int64_t)k.KS * 64;
    for (int mt0 = 0; mt0 < k.MT; mt0 += MTW) {
        f32x16 acc[MTW][4];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][v][r] = 0.f;
        const float* wp = wbase + (int64_t)mt0 * mt_stride;
        // No per-k-step branch (each would be its own basic block with the weight load pinned in front of its MFMAs);
        // steps beyond KS run on zero operands.  Weights arrive in batches of AB k-steps, one batch ahead.
        constexpr int AB = 4;
        static_assert(KSM % AB == 0, "KSM must be a multiple of the weight batch");
        auto load_w = [&](int sb, float (&dst)[AB][MTW]) {
#pragma unroll
            for (int u = 0; u < AB; ++u)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    const float w = wp[(mt0 + m < k.MT ? m : 0) * mt_stride + (int64_t)min(sb + u, k.KS - 1) * 64];   // valid address
                    dst[u][m] = w * ((sb + u < k.KS && mt0 + m < k.MT) ? 1.f : 0.f);    // multiply, not select (see stream kernel)
                }
        };
        float avn[AB][MTW];
        load_w(0, avn);
        if (!(k.dbg & 1))
#pragma unroll
        for (int sb = 0; sb < KSM; sb += AB) {
            float avc[AB][MTW];
#pragma unroll
            for (int u = 0; u < AB; ++u)
#pragma unroll
                for (int m = 0; m < MTW; ++m) avc[u][m] = avn[u][m];
            if (sb + AB < KSM) load_w(sb + AB, avn);
#pragma unroll
            for (int u = 0; u < AB; ++u)
#pragma unroll
                for (int m = 0; m < MTW; ++m) {
                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[u][m], xr[sb + u].x, acc[m][0], 0, 0, 0);
                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[u][m], xr[sb + u].y, acc[m][1], 0, 0, 0);
                    acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[u][m], xr[sb + u].z, acc[m][2], 0, 0, 0);
                    acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(avc[u][m], xr[sb + u].w, acc[m][3], 0, 0, 0);
                }
        }
        if (any && !(k.dbg & 2)) pw3_epilogue<MTW>(k, b, mt0, p, vec, half, acc);
        if (k.dbg & 2) { float sacc = 0.f;
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int v = 0; v < 4; ++v) sacc += acc[m][v][0] + acc[m][v][7] + acc[m][v][15];
            if (sacc == 123.456f) k.out[0] = sacc; }
    }
}

template <int MTW, bool VEC, bool SUM>
__global__ __launch_bounds__(256, 2) void pw_gemm3_stream_kernel(PwK k) {
    constexpr int PF = 4;   // k-steps per batch
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int b = blockIdx.z;
    const int mt0 = blockIdx.y * MTW;
    const int p = (blockIdx.x * 4 + wave) * 128 + 4 * (lane & 31);
    if (p - 4 * (lane & 31) >= k.L) return;
    const bool vec = (k.L % 4 == 0) && (p + 3 < k.L);
    const bool any = p < k.L;
    f32x16 acc[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][v][r] = 0.f;
    const float* wp = k.Wp + (int64_t)b * k.w_bstride + ((int64_t)mt0 * k.KS) * 64 + lane;
    const int64_t mt_stride = (int64_t)k.KS * 64;
    auto load_batch = [&](int s0, float4 (&xb)[PF], float (&ab)[MTW][PF]) {
        load_steps<PF, VEC, SUM>(k, b, s0, half, p, any, xb);       // branch-free; steps past K give zeros
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int sn = s0 + u;
            const int sc = min(sn, k.KS - 1);                        // clamped address, value selected
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                // valid address always; masked by MULTIPLYING: a select on this wave-uniform condition is turned back into a
                // branch around the load by hipcc, and every such block ends in s_waitcnt vmcnt(0)
                const float w = wp[(mt0 + m < k.MT ? m : 0) * mt_stride + (int64_t)sc * 64];
                ab[m][u] = w * ((mt0 + m < k.MT && sn < k.KS) ? 1.f : 0.f);
            }
        }
    };
    auto mma_batch = [&](int s0, const float4 (&xb)[PF], const float (&ab)[MTW][PF]) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {      // steps beyond KS carry zero operands: no branch per step
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[m][u], xb[u].x, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[m][u], xb[u].y, acc[m][1], 0, 0, 0);
                acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[m][u], xb[u].z, acc[m][2], 0, 0, 0);
                acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[m][u], xb[u].w, acc[m][3], 0, 0, 0);
            }
        }
    };
    // three register batches rotate: two are always in flight behind the one being multiplied (one batch = 4 k-steps =
    // 32 MFMAs ~ 0.9 us of matrix-core time, less than a loaded HBM round trip, so a single batch ahead still stalls)
    float4 xA[PF], xB[PF], xC[PF];
    float aA[MTW][PF], aB[MTW][PF], aC[MTW][PF];
    load_batch(0, xA, aA);
    load_batch(PF, xB, aB);
    for (int s0 = 0; s0 < k.KS; s0 += 3 * PF) {
        load_batch(s0 + 2 * PF, xC, aC);
        mma_batch(s0, xA, aA);
        load_batch(s0 + 3 * PF, xA, aA);
        mma_batch(s0 + PF, xB, aB);
        load_batch(s0 + 4 * PF, xB, aB);
        mma_batch(s0 + 2 * PF, xC, aC);
    }
    if (any) pw3_epilogue<MTW>(k, b, mt0, p, vec, half, acc);
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
    static int pw_dbg = getenv("BEM_PW_DBG") ? atoi(getenv("BEM_PW_DBG")) : 0;
    k.dbg = pw_dbg;
    hipStream_t s = (hipStream_t)stream;
    const int Kp = 2 * k.KS;
    const bool ln = a->ln_w != nullptr;
    static int force_v1 = getenv("BEM_PW_V1") ? atoi(getenv("BEM_PW_V1")) : 0;   // 1: v1 only, 2: v2/v1 only
    if (!force_v1) {
        // v3: barrier-free.  Register-resident input for K <= 80 (LayerNorm or not), software-pipelined stream otherwise.
        dim3 grid(cdiv(a->L, 512), 1, a->B);
        const bool vecL = (a->L % 4 == 0), sum = (a->in_mode == 1);
#define BEM_PW_VS(KERNEL, ...)                                                     \
    do {                                                                           \
        if (vecL && !sum) KERNEL<__VA_ARGS__, true, false><<<grid, 256, 0, s>>>(k);     \
        else if (vecL && sum) KERNEL<__VA_ARGS__, true, true><<<grid, 256, 0, s>>>(k); \
        else if (!sum) KERNEL<__VA_ARGS__, false, false><<<grid, 256, 0, s>>>(k);       \
        else KERNEL<__VA_ARGS__, false, true><<<grid, 256, 0, s>>>(k);                  \
    } while (0)
        if (k.KS <= 20) {
            if (k.MT == 1) BEM_PW_VS(pw_gemm3_reg_kernel, 20, 1);
            else BEM_PW_VS(pw_gemm3_reg_kernel, 20, 2);
            return bem_check_launch("pw_gemm3_reg");
        }
        if (!ln) {
            if (k.MT == 1) {
                BEM_PW_VS(pw_gemm3_stream_kernel, 1);
            } else {
                grid.y = cdiv(k.MT, 2);
                BEM_PW_VS(pw_gemm3_stream_kernel, 2);
            }
            return bem_check_launch("pw_gemm3_stream");
        }
#undef BEM_PW_VS
    }
    if (force_v1 != 1 && !(ln && Kp > 160)) {
        // v2: LDS-staged.  LayerNorm needs the whole K in LDS (<= 160 channels = 80 KiB); otherwise stream KCH-channel chunks.
        const int kchunk = ln ? Kp : (Kp < KCH ? Kp : KCH);
        const size_t lds = ((size_t)kchunk * PT + 2 * PT) * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_gemm2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_set = true;
        }
        dim3 grid(cdiv(a->L, PT), a->B);
        if (k.MT >= 4)
            pw_gemm2_kernel<true><<<grid, 256, lds, s>>>(k, kchunk);
        else
            pw_gemm2_kernel<false><<<grid, 256, lds, s>>>(k, kchunk);
        return bem_check_launch("pw_gemm2");
    }
    if (k.MT == 1) {
        dim3 grid(cdiv(a->L, 512), 1, a->B);
        pw_gemm_kernel<1><<<grid, 256, 0, s>>>(k);
    } else {
        dim3 grid(cdiv(a->L, 512), cdiv(k.MT, 2), a->B);
        pw_gemm_kernel<2><<<grid, 256, 0, s>>>(k);
    }
    return bem_check_launch("pw_gemm");
}
