// Pointwise (1x1) channel-mix GEMM, f32 in / f32 out, products on the bf16 matrix cores as a 3-limb expansion.
//
//   out[b][m][p] = act(sum_k W[m][k] * pro(x)[b][k][p] + bias[m]) + res[b][m][p]        (argument block bem_pw_args, include/bem_hip.h)
//
// Why: v_mfma_f32_32x32x2_f32 delivers 64 FLOP / cycle / SIMD, v_mfma_f32_32x32x16_bf16 1024.  Every f32 operand is
// split exactly into three bf16 limbs  v = h + m + l  (round-to-nearest at each step, exact f32 residuals), and a
// product is evaluated as the six limb products of weight >= 2^-16 relative
//        x*w ~ xl*wh + xh*wl + xm*wm + xm*wh + xh*wm + xh*wh          (dropped: xm*wl + xl*wm + xl*wl <= 2^-23 |x w|)
// each of which is exact in the f32 accumulator of the MFMA.  The result carries f32-level error (the dropped terms
// are of the order of one f32 rounding of the product) at 6/16 of the f32-MFMA cost, which moves these GEMMs from
// the matrix pipe to the HBM roofline.  Parity tests hold it to the same tolerances as the native-f32 kernels.
//
// Mapping (NCHW, pixels on lanes):
//   * one wave = 32 * NSUB consecutive pixels; lane l owns pixels p0 + NSUB * (l & 31) + t, t < NSUB (one float2 load
//     per lane and channel for NSUB = 2).  Sub-tile t (pixels with the same t) is one MFMA N-tile.
//   * k-block kb covers input channels 16 kb .. 16 kb + 15; lanes 0-31 hold channels 16 kb + 0..7, lanes 32-63
//     channels 16 kb + 8..15 (the B[k = 8 (l >> 5) + e][n = l & 31] layout of the 32x32x16 instruction).
//   * weights arrive pre-split and pre-packed by bem_pack_pw_weight_x6: Wp[mtile][kb][limb][lane] is one 16-byte
//     vector = W[32 mtile + (lane & 31)][16 kb + 8 (lane >> 5) + e], e = 0..7, so an A operand is one coalesced
//     1 KiB load per (M-tile, k-block, limb), L1/L2 resident.
//   * D layout: column = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
//   * "resident" kernel: the wave's whole input tile (all K channels, already normalised and split) stays in
//     registers while the wave walks over every M-tile, so x is read once and LayerNorm evaluated once;
//     "stream" kernel (no LayerNorm, any K): x is streamed k-block by k-block, one block ahead of the MFMAs.
#include "bem_common.h"
#include "x6_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {


struct PwX {
    const float* x1; const float* x2; int C1; int C2; int in_mode;
    const float* ln_w; const float* ln_b; float ln_eps;
    const u32x4* Wp; int64_t w_bstride;       // stride in 16-byte vectors
    const float* bias; int64_t bias_bstride;
    const float* res; const float* prelu; int act;
    float* out; int out_mode; int Win;
    int M; int K; int L; int KB; int MT;
    int mtpb;      // resident kernel: M-tiles per workgroup (grid.y slices M when there are too few pixels to fill the GPU)
};


// Branch-free input fetch: always a clamped, valid address, value masked afterwards.
// Returns pro-input channel ch at this lane's NSUB pixels (pixel index pc clamped by the caller, keep[t] per pixel).
template <int NSUB, bool SUM, bool VEC>
__device__ __forceinline__ void ldx(const PwX& k, int b, int ch, int pc, const bool (&keep)[NSUB], float (&o)[NSUB]) {
    const int chc = min(ch, k.K - 1);
    const bool first = chc < k.C1;
    const float* r1 = k.x1 + ((int64_t)b * k.C1 + (first ? chc : 0)) * k.L;
    const float* r2 = k.x2 + ((int64_t)b * k.C2 + (first ? 0 : chc - k.C1)) * k.L;
    const float* base = first ? r1 : r2;
    const float* sec = k.x2 + ((int64_t)b * k.C2 + chc) * k.L;     // only dereferenced when SUM
    if (NSUB == 2 && VEC) {
        float2 v = *reinterpret_cast<const float2*>(base + pc);
        if (SUM) { const float2 w = *reinterpret_cast<const float2*>(sec + pc); v.x += w.x; v.y += w.y; }
        o[0] = v.x; o[NSUB - 1] = v.y;
    } else {
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            const int pi = min(pc + t, k.L - 1);
            o[t] = base[pi];
            if (SUM) o[t] += sec[pi];
        }
    }
    const float mk = ch < k.K ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NSUB; ++t) o[t] = keep[t] ? o[t] * mk : 0.f;
}

// bias comes from LDS (s_bias, zero-filled without a bias), the residual is fetched under ONE uniform branch per batch
// of rows: every global round trip in here is latency the wave cannot hide (2 waves per SIMD), so there is none
// unless a residual is actually present.
template <int MTW, int NSUB, bool VEC>
__device__ __forceinline__ void x6_epilogue_generic(const PwX& k, int b, int mt0, int p, const bool (&keep)[NSUB], int kh,
                                                    const float* __restrict__ s_bias, const f32x16 (&acc)[MTW][NSUB]) {
    const float slope = (k.act == 1) ? k.prelu[0] : 0.f;
    const bool has_res = k.res && k.out_mode == 0;
    const float* resp = k.res + (int64_t)b * k.M * k.L;
    const int pv = VEC ? (keep[0] ? p : 0) : p;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (mt0 + m >= k.MT) continue;
        const int rbase = (mt0 + m) * 32 + 4 * kh;
#pragma unroll
        for (int rh = 0; rh < 16; rh += 8) {
            float rv[8][NSUB];
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8)
#pragma unroll
                for (int t = 0; t < NSUB; ++t) rv[r8][t] = 0.f;
            if (has_res) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = rh + r8;
                    const float* rp = resp + (int64_t)min(rbase + (r & 3) + 8 * (r >> 2), k.M - 1) * k.L;
                    if (NSUB == 2 && VEC) {
                        const float2 q = *reinterpret_cast<const float2*>(rp + pv);
                        rv[r8][0] = q.x; rv[r8][NSUB - 1] = q.y;
                    } else {
#pragma unroll
                        for (int t = 0; t < NSUB; ++t) rv[r8][t] = rp[min(pv + t, k.L - 1)];
                    }
                }
            }
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = rh + r8;
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row >= k.M) continue;
                const float bv = s_bias[row];
                float o[NSUB];
#pragma unroll
                for (int t = 0; t < NSUB; ++t) {
                    o[t] = acc[m][t][r] + bv;
                    if (k.act == 1) o[t] = o[t] >= 0.f ? o[t] : slope * o[t];
                    o[t] += rv[r8][t];
                }
                if (k.out_mode == 0) {
                    float* op = k.out + ((int64_t)b * k.M + row) * k.L + p;
                    if (NSUB == 2 && VEC) {
                        if (keep[0]) *reinterpret_cast<float2*>(op) = make_float2(o[0], o[NSUB - 1]);
                    } else {
#pragma unroll
                        for (int t = 0; t < NSUB; ++t)
                            if (keep[t]) op[t] = o[t];
                    }
                } else {
                    const int Co = k.M >> 2;
                    const int q = row / Co, co = row - q * Co;
                    const int64_t obase = ((int64_t)b * Co + co) * (4 * (int64_t)k.L);
#pragma unroll
                    for (int t = 0; t < NSUB; ++t) {
                        if (keep[t]) {
                            const int pp = p + t;
                            const int yy = pp / k.Win, xx = pp - yy * k.Win;
                            k.out[obase + (int64_t)(2 * yy + (q >> 1)) * (2 * k.Win) + (2 * xx + (q & 1))] = o[t];
                        }
                    }
                }
            }
        }
    }
}

// out_mode 0 epilogue with the address arithmetic kept off the vector ALU: a row's plane base is wave-uniform (scalar
// registers), the lane contributes one 32-bit element offset computed once (its pixel and its 4-row half), so a store
// is `global_store v_off, v_data, s[base]`; the 16 bias values of an M-tile come as four LDS float4 reads.  PReLU and
// the residual are compile-time variants (chosen by uniform branches in x6_epilogue): without them a value costs one add.
template <int MTW, int NSUB, bool VEC, bool ACT, bool RES>
__device__ __forceinline__ void x6_epilogue_rows(const PwX& k, int b, int mt0, int p, const bool (&keep)[NSUB], int kh, const float* s_bias,
                                                 const float4 (&bq)[MTW][4], const f32x16 (&acc)[MTW][NSUB]) {
    const float slope = ACT ? k.prelu[0] : 0.f;
    const int pv = VEC ? (keep[0] ? p : 0) : min(p, k.L - 1);
    const uint32_t loff = (uint32_t)(4 * kh) * (uint32_t)k.L + (uint32_t)pv;
    float* outb = k.out + (int64_t)b * k.M * k.L;
    const float* resb = RES ? k.res + (int64_t)b * k.M * k.L : nullptr;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (mt0 + m >= k.MT) continue;
        const int rb = (mt0 + m) * 32;                        // uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#ifdef BEM_X6_BIAS_LDS      // A/B diagnostic builds only (make dbg, scripts/x6_bias_ab.py): bias read back from LDS
#if BEM_X6_BIAS_LDS == 2      // 2: four separate dword reads, all retired (lgkmcnt(0)) before the first use
            const float* sb = s_bias + rb + 8 * g + 4 * kh;
            const float4 b4 = make_float4(sb[0], sb[1], sb[2], sb[3]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
#elif BEM_X6_BIAS_LDS >= 3    // 3: one 16-byte read retired before the first use (the builtin keeps the compiler from sinking components); 4: + scalar adds
            const float4 b4 = *reinterpret_cast<const float4*>(s_bias + rb + 8 * g + 4 * kh);
            __builtin_amdgcn_s_waitcnt(0xc07f);
#else                         // 1: the round-1 form: the compiler sinks the components into the conditional row blocks as
                              //    ds_read2_b32 + 2 x ds_read_b32 and waits for them with COUNTED lgkmcnt(1) / lgkmcnt(0)
            const float4 b4 = *reinterpret_cast<const float4*>(s_bias + rb + 8 * g + 4 * kh);
#endif
            const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#else
            // through a VALU copy: a packed add that takes the HIGH register of a freshly loaded pair for its LOW result read the pair's
            // pre-load content in lanes 48..63 (round 3: the M = 32, K = 64 attention-fuse GEMM at 280 workgroups, scripts/dbg_gemm_cat.py;
            // round 2 saw the same with LDS-loaded pairs) -- DESIGN.md section 6.4, scripts/isa_audit.py check 2
            const float bv[4] = {valu_copy(bq[m][g].x), valu_copy(bq[m][g].y), valu_copy(bq[m][g].z), valu_copy(bq[m][g].w)};
#endif
            float rv[4][NSUB];
            if (RES) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int urow = rb + 8 * g + i;
                    const float* rp = resb + (int64_t)min(urow, k.M - 1) * k.L;          // uniform plane base, clamped
                    const uint32_t lo = urow + 4 * kh < k.M ? loff : (uint32_t)pv;        // rows >= M (never stored) read a valid row
                    if (NSUB == 2 && VEC) {
                        const float2 q = *reinterpret_cast<const float2*>(rp + lo);
                        rv[i][0] = q.x; rv[i][NSUB - 1] = q.y;
                    } else {
#pragma unroll
                        for (int t = 0; t < NSUB; ++t) rv[i][t] = rp[lo + (pv + t < k.L ? t : 0)];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i;
                const int urow = rb + 8 * g + i;              // uniform; this lane's row = urow + 4 kh
                float o[NSUB];
#pragma unroll
                for (int t = 0; t < NSUB; ++t) {
                    o[t] = acc[m][t][r] + bv[i];
#if defined(BEM_X6_BIAS_LDS) && BEM_X6_BIAS_LDS == 4    // diagnostic: keep the two adds of a row scalar (no v_pk_add_f32)
                    asm volatile("" : "+v"(o[t]));
#endif
                    if (ACT) o[t] = o[t] >= 0.f ? o[t] : slope * o[t];
                    if (RES) o[t] += rv[i][t];
                }
                float* op = outb + (int64_t)urow * k.L;
                const bool rowok = urow + 4 * kh < k.M;
                if (NSUB == 2 && VEC) {
                    if (keep[0] && rowok) *reinterpret_cast<float2*>(op + loff) = make_float2(o[0], o[NSUB - 1]);
                } else {
#pragma unroll
                    for (int t = 0; t < NSUB; ++t)
                        if (keep[t] && rowok) op[loff + t] = o[t];
                }
            }
        }
    }
}

// This lane's 16 bias values of each M-tile of a group (rows rb + 8 g + 4 kh + i), fetched with the group's first weights
// so that the epilogue finds them in registers.  M % 4 == 0: four 16-byte loads (the half-wave reads one address);
// otherwise scalar reads with clamped rows.  (An LDS copy read back as float4 returned a stale third component for the
// upper half-wave a few times per 10^7 outputs under load; it is used by the generic epilogue only, as scalars.)
template <int MTW>
__device__ __forceinline__ void x6_load_bias(const PwX& k, int b, int mt0, int kh, float4 (&bq)[MTW][4]) {
    const float* gb = k.bias ? k.bias + (int64_t)b * k.bias_bstride : nullptr;      // uniform
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[m][g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!gb) return;
    if ((k.M & 3) == 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bq[m][g] = *reinterpret_cast<const float4*>(gb + min((mt0 + m) * 32 + 8 * g + 4 * kh, k.M - 4));   // rows >= M are never stored
    } else {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r0 = (mt0 + m) * 32 + 8 * g + 4 * kh;
                bq[m][g] = make_float4(gb[min(r0, k.M - 1)], gb[min(r0 + 1, k.M - 1)], gb[min(r0 + 2, k.M - 1)], gb[min(r0 + 3, k.M - 1)]);
            }
    }
}

template <int MTW, int NSUB, bool VEC>
__device__ __forceinline__ void x6_epilogue(const PwX& k, int b, int mt0, int p, const bool (&keep)[NSUB], int kh,
                                            const float* __restrict__ s_bias, const float4 (&bq)[MTW][4], const f32x16 (&acc)[MTW][NSUB]) {
    if (k.out_mode != 0 || k.M < 8) { x6_epilogue_generic<MTW, NSUB, VEC>(k, b, mt0, p, keep, kh, s_bias, acc); return; }
    const bool act = k.act == 1, res = k.res != nullptr;      // uniform
    if (!act && !res) x6_epilogue_rows<MTW, NSUB, VEC, false, false>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
    else if (!act) x6_epilogue_rows<MTW, NSUB, VEC, false, true>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
    else if (!res) x6_epilogue_rows<MTW, NSUB, VEC, true, false>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
    else x6_epilogue_rows<MTW, NSUB, VEC, true, true>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
}

// bias of this batch row -> LDS (zeros without a bias); M <= BEM_X6_MAXM
constexpr int BEM_X6_MAXM = 2048;
__device__ __forceinline__ void stage_bias(const PwX& k, int b, float* s_bias) {
    const float* bias = k.bias ? k.bias + (int64_t)b * k.bias_bstride : nullptr;
    for (int i = threadIdx.x; i < k.MT * 32; i += 256) s_bias[i] = (bias && i < k.M) ? bias[i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// resident: K <= 16 * KBM.  grid (ceil(L / (128 * NSUB)), 1, B), 4 independent waves per workgroup.
// ------------------------------------------------------------------------------------------------
template <int KBM, int NSUB, int MTW, bool SUM, bool VEC>
__global__ __launch_bounds__(256, 2) void pw_x6_res_kernel(PwX k) {
    __shared__ float s_ln[2 * 16 * KBM];
    __shared__ __attribute__((aligned(16))) float s_bias[BEM_X6_MAXM];
    stage_bias(k, blockIdx.z, s_bias);
    for (int i = threadIdx.x; i < 16 * KBM; i += 256) {
        const bool on = k.ln_w && i < k.K;
        s_ln[i] = on ? k.ln_w[min(i, k.K - 1)] : 0.f;
        s_ln[16 * KBM + i] = on ? k.ln_b[min(i, k.K - 1)] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z;
    const int p0 = (xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * (32 * NSUB);
    const int p = p0 + NSUB * n;
    bool keep[NSUB];
#pragma unroll
    for (int t = 0; t < NSUB; ++t) keep[t] = p + t < k.L;
    // VEC (L even, p even): both pixels are kept or neither -> read at p or at 0; scalar form: ldx clamps every pixel itself
    const int pc = VEC ? (keep[0] ? p : 0) : p;
    float xr[KBM][8][NSUB];
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) ldx<NSUB, SUM, VEC>(k, b, 16 * kb + 8 * kh + e, pc, keep, xr[kb][e]);
    __syncthreads();                                            // s_ln visible (the only barrier; before any early exit)
    if (p0 >= k.L) return;
    if (k.ln_w) {
        const float inv = 1.f / (float)k.K;
        float mean[NSUB], rstd[NSUB];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            float s = 0.f;
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xr[kb][e][t];
            s += __shfl_xor(s, 32, 64);
            mean[t] = s * inv;
            float q = 0.f;
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (16 * kb + 8 * kh + e < k.K) ? xr[kb][e][t] - mean[t] : 0.f;     // padded channels do not count
                    q = fmaf(d, d, q);
                }
            q += __shfl_xor(q, 32, 64);
            rstd[t] = 1.f / sqrtf(q * inv + k.ln_eps);
        }
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // through a VALU copy: packed-f32 ops must not consume LDS-returned register pairs directly
                const float g = valu_copy(s_ln[16 * kb + 8 * kh + e]), be = valu_copy(s_ln[16 * KBM + 16 * kb + 8 * kh + e]);   // 0 on padded channels
#pragma unroll
                for (int t = 0; t < NSUB; ++t) xr[kb][e][t] = (xr[kb][e][t] - mean[t]) * rstd[t] * g + be;
            }
    }
    u32x4 xl[KBM][NSUB][3];
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xr[kb][e][t];
            split8(v, xl[kb][t][0], xl[kb][t][1], xl[kb][t][2]);
        }
    const u32x4* wbase = k.Wp + (int64_t)b * k.w_bstride + lane;
    const int64_t mt_stride = (int64_t)k.KB * 3 * 64;           // vectors per M-tile
    // weights one k-block ahead of the MFMAs, across M-tile groups too (the first block of the next group is requested
    // before the epilogue of this one); k-blocks beyond KB and M-tiles beyond MT re-read a valid block and are masked to zero
    auto load_w = [&](int mt, int kb, u32x4 (&dst)[MTW][3]) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const bool ok = kb < k.KB && mt + m < k.MT;
            const u32x4* wp = wbase + (int64_t)(mt + m < k.MT ? mt + m : 0) * mt_stride + (int64_t)min(kb, k.KB - 1) * 3 * 64;
            const uint32_t mk = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int li = 0; li < 3; ++li) {
                const u32x4 w = wp[li * 64];
                dst[m][li] = u32x4{w[0] & mk, w[1] & mk, w[2] & mk, w[3] & mk};
            }
        }
    };
    u32x4 wn[MTW][3];
    const int mt_lo = blockIdx.y * k.mtpb, mt_hi = min(k.MT, mt_lo + k.mtpb);
    load_w(mt_lo, 0, wn);
    for (int mt0 = mt_lo; mt0 < mt_hi; mt0 += MTW) {
        f32x16 acc[MTW][NSUB], alo[MTW][NSUB];
        float4 bq[MTW][4];
        x6_load_bias<MTW>(k, b, mt0, kh, bq);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NSUB; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][t][r] = alo[m][t][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            u32x4 wc[MTW][3];
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int li = 0; li < 3; ++li) wc[m][li] = wn[m][li];
            if (kb + 1 < KBM) load_w(mt0, kb + 1, wn);
            else load_w(mt0 + MTW, 0, wn);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int t = 0; t < NSUB; ++t) mac6(wc[m], xl[kb][t], acc[m][t], alo[m][t]);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NSUB; ++t) acc[m][t] += alo[m][t];
        x6_epilogue<MTW, NSUB, VEC>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// resident, weights through LDS: K = 16 * KBM exactly, many M-tiles (level-2 project_in: K 160 -> M 1280).  In pw_x6_res_kernel each
// of the four waves pulls every weight block from L2 for its own 32 pixels, one k-block ahead: 1.2 MB per wave at two waves per SIMD is a
// latency chain (7 TB/s of L2 reads in flight-limited pieces, matrix pipe 15 % busy).  Here the workgroup fetches an M-tile's KBM * 3 KiB of
// weights ONCE by LDS-DMA, a whole M-tile ahead of the MFMAs that read it, and the four waves share it: a quarter of the L2 traffic and
// 30 KiB in flight per workgroup.  One barrier per M-tile; waves beyond the image keep running (masked stores) so that every wave reaches it.
// ------------------------------------------------------------------------------------------------
template <int KBM, bool SUM, bool VEC>
__global__ __launch_bounds__(256, 2) void pw_x6_res_lds_kernel(PwX k) {
    __shared__ float s_ln[2 * 16 * KBM];
    __shared__ __attribute__((aligned(16))) float s_bias[BEM_X6_MAXM];
    __shared__ __attribute__((aligned(16))) u32x4 Ws[2][KBM * 3 * 64];
    stage_bias(k, blockIdx.z, s_bias);
    for (int i = threadIdx.x; i < 16 * KBM; i += 256) {
        const bool on = k.ln_w && i < k.K;
        s_ln[i] = on ? k.ln_w[min(i, k.K - 1)] : 0.f;
        s_ln[16 * KBM + i] = on ? k.ln_b[min(i, k.K - 1)] : 0.f;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z;
    const int p0 = (xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * 32;
    const int p = p0 + n;
    bool keep[1] = {p < k.L};
    const int pc = VEC ? (keep[0] ? p : 0) : p;
    const int mt_lo = blockIdx.y * k.mtpb, mt_hi = min(k.MT, mt_lo + k.mtpb);
    const u32x4* wsrc = k.Wp + (int64_t)b * k.w_bstride;
    const int64_t mt_stride = (int64_t)KBM * 3 * 64;
    auto dma = [&](int mt, int buf) {                               // KBM * 3 pieces of 1 KiB, dealt round-robin to the four waves
        const u32x4* src = wsrc + (int64_t)mt * mt_stride;
        for (int piece = wave; piece < KBM * 3; piece += 4)
            glds16(src + piece * 64, (uint32_t)lane * 16u, lds_addr(&Ws[buf][piece * 64]));
    };
    dma(mt_lo, 0);
    float xr[KBM][8][1];
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) ldx<1, SUM, VEC>(k, b, 16 * kb + 8 * kh + e, pc, keep, xr[kb][e]);
    __syncthreads();                                                // s_ln, s_bias visible
    if (k.ln_w) {
        const float inv = 1.f / (float)k.K;
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xr[kb][e][0];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * inv;
        float q = 0.f;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xr[kb][e][0] - mean;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.f / sqrtf(q * inv + k.ln_eps);
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = valu_copy(s_ln[16 * kb + 8 * kh + e]), be = valu_copy(s_ln[16 * KBM + 16 * kb + 8 * kh + e]);
                xr[kb][e][0] = (xr[kb][e][0] - mean) * rstd * g + be;
            }
    }
    u32x4 xl[KBM][3];
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = xr[kb][e][0];
        split8(v, xl[kb][0], xl[kb][1], xl[kb][2]);
    }
    int buf = 0;
    for (int mt0 = mt_lo; mt0 < mt_hi; ++mt0, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of M-tile mt0 have landed ...
        __syncthreads();                                            // ... everybody's have, and everybody is done reading the other buffer
        if (mt0 + 1 < mt_hi) dma(mt0 + 1, buf ^ 1);
        f32x16 acc[1][1], alo;
        float4 bq[1][4];
        x6_load_bias<1>(k, b, mt0, kh, bq);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = alo[r] = 0.f;
        const u32x4* wl = &Ws[buf][lane];
        u32x4 wn[3] = {wl[0], wl[64], wl[128]};
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {                         // LDS reads one k-block ahead of the MFMAs, no further (register budget)
            const u32x4 wc[3] = {wn[0], wn[1], wn[2]};
            if (kb + 1 < KBM) {
                wn[0] = wl[((kb + 1) * 3 + 0) * 64]; wn[1] = wl[((kb + 1) * 3 + 1) * 64]; wn[2] = wl[((kb + 1) * 3 + 2) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            mac6(wc, xl[kb], acc[0][0], alo);
            __builtin_amdgcn_sched_barrier(0);
        }
        acc[0][0] += alo;
        x6_epilogue<1, 1, VEC>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// stream: any K, no LayerNorm.  grid (ceil(L / 256), ceil(MT / MTW), B); x one k-block ahead of the MFMAs.
// ------------------------------------------------------------------------------------------------
constexpr int BEM_X6_MAXK_LN = 1024;      // LayerNorm parameters of the streaming form live in LDS
template <int MTW, int NSUB, bool SUM, bool VEC, bool LN>
__global__ __launch_bounds__(256, 2) void pw_x6_stream_kernel(PwX k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z, mt0 = blockIdx.y * MTW;
    const int p0 = (xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * (32 * NSUB);
    __shared__ __attribute__((aligned(16))) float s_bias[BEM_X6_MAXM];
    __shared__ float s_ln[LN ? 2 * BEM_X6_MAXK_LN : 2];
    stage_bias(k, b, s_bias);
    if (LN) {
        for (int i = threadIdx.x; i < k.KB * 16; i += 256) {     // zero scale / shift on the padded channels
            s_ln[i] = i < k.K ? k.ln_w[i] : 0.f;
            s_ln[BEM_X6_MAXK_LN + i] = i < k.K ? k.ln_b[i] : 0.f;
        }
    }
    __syncthreads();                    // the only barrier, before any early exit
    if (p0 >= k.L) return;
    const int p = p0 + NSUB * n;
    bool keep[NSUB];
#pragma unroll
    for (int t = 0; t < NSUB; ++t) keep[t] = p + t < k.L;
    const int pc = VEC ? (keep[0] ? p : 0) : p;
    auto load_x = [&](int kb, float (&dst)[8][NSUB]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ldx<NSUB, SUM, VEC>(k, b, 16 * kb + 8 * kh + e, pc, keep, dst[e]);   // channels >= K come back as zeros
    };
    // LayerNorm over K larger than the register-resident forms hold: statistics in a first sweep over the channels
    // (mean, then centred variance -- the second and third reads of x come from L1 / L2), normalisation on the fly below
    float mean[NSUB], rstd[NSUB];
    if (LN) {
        const float inv = 1.f / (float)k.K;
        float s[NSUB];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) s[t] = 0.f;
        for (int kb = 0; kb < k.KB; ++kb) {
            float v[8][NSUB];
            load_x(kb, v);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int t = 0; t < NSUB; ++t) s[t] += v[e][t];
        }
#pragma unroll
        for (int t = 0; t < NSUB; ++t) { s[t] += __shfl_xor(s[t], 32, 64); mean[t] = s[t] * inv; }
        float q[NSUB];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) q[t] = 0.f;
        for (int kb = 0; kb < k.KB; ++kb) {
            float v[8][NSUB];
            load_x(kb, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mk = (16 * kb + 8 * kh + e < k.K) ? 1.f : 0.f;
#pragma unroll
                for (int t = 0; t < NSUB; ++t) { const float d = (v[e][t] - mean[t]) * mk; q[t] = fmaf(d, d, q[t]); }
            }
        }
#pragma unroll
        for (int t = 0; t < NSUB; ++t) { q[t] += __shfl_xor(q[t], 32, 64); rstd[t] = 1.f / sqrtf(q[t] * inv + k.ln_eps); }
    }
    f32x16 acc[MTW][NSUB], alo[MTW][NSUB];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NSUB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = alo[m][t][r] = 0.f;
    const u32x4* wbase = k.Wp + (int64_t)b * k.w_bstride + lane;
    const int64_t mt_stride = (int64_t)k.KB * 3 * 64;
    auto load_w = [&](int kb, u32x4 (&dst)[MTW][3]) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const bool ok = kb < k.KB && mt0 + m < k.MT;
            const u32x4* wp = wbase + (int64_t)(mt0 + m < k.MT ? mt0 + m : 0) * mt_stride + (int64_t)min(kb, k.KB - 1) * 3 * 64;
            const uint32_t mk = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int li = 0; li < 3; ++li) {
                const u32x4 w = wp[li * 64];
                dst[m][li] = u32x4{w[0] & mk, w[1] & mk, w[2] & mk, w[3] & mk};
            }
        }
    };
    // x two k-blocks ahead (the HBM stream: with 2 waves per SIMD one block in flight per wave covers ~4 TB/s at 2 us of
    // latency), weights one ahead (L1 / L2)
    float xn[8][NSUB], xn2[8][NSUB];
    u32x4 wn[MTW][3];
    load_x(0, xn);
    load_x(1, xn2);
    load_w(0, wn);
    for (int kb = 0; kb < k.KB; ++kb) {
        u32x4 xl[NSUB][3], wc[MTW][3];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = xn[e][t];
                if (LN) v[e] = (v[e] - mean[t]) * rstd[t] * valu_copy(s_ln[16 * kb + 8 * kh + e]) + valu_copy(s_ln[BEM_X6_MAXK_LN + 16 * kb + 8 * kh + e]);
            }
            split8(v, xl[t][0], xl[t][1], xl[t][2]);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int li = 0; li < 3; ++li) wc[m][li] = wn[m][li];
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < NSUB; ++t) xn[e][t] = xn2[e][t];
        load_x(kb + 2, xn2);         // past the end: clamped channel, masked to zero, never used
        load_w(kb + 1, wn);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NSUB; ++t) mac6(wc[m], xl[t], acc[m][t], alo[m][t]);
    }
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NSUB; ++t) acc[m][t] += alo[m][t];
    float4 bq[MTW][4];                  // after the k loop: 16 more live registers per M-tile inside it would spill
    x6_load_bias<MTW>(k, b, mt0, kh, bq);
    x6_epilogue<MTW, NSUB, VEC>(k, b, mt0, p, keep, kh, s_bias, bq, acc);
}

// natural (nsets, M, K) f32 -> (nsets, MT, KB, 3, 64) 16-byte vectors of bf16 limbs
__device__ __forceinline__ void pack_x6_item(int64_t i, const float* __restrict__ W, u32x4* __restrict__ Wp, int M, int K, int MT, int KB,
                                             int64_t ss, int64_t rs, int64_t cs) {      // element strides of W over (set, row, k): transposed / sliced views pack in place
    const int lane = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kb = (int)(blk % KB), mt = (int)((blk / KB) % MT);
    const int64_t set = blk / ((int64_t)KB * MT);
    const int row = mt * 32 + (lane & 31), k0 = kb * 16 + (lane >> 5) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (row < M && k0 + e < K) ? W[set * ss + row * rs + (k0 + e) * cs] : 0.f;
    u32x4 h, m, l;
    split8(v, h, m, l);
    u32x4* o = Wp + ((set * MT + mt) * KB + kb) * 3 * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

__global__ void pack_x6_kernel(const float* __restrict__ W, u32x4* __restrict__ Wp, int M, int K, int MT, int KB, int64_t total,
                               int64_t ss, int64_t rs, int64_t cs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    pack_x6_item(i, W, Wp, M, K, MT, KB, ss, rs, cs);
}

// Many (M, K) matrices packed by one launch: job = { source pointer, first float of the packed output in `arena` (multiple of 4), int32 M,
// int32 K, row stride, column stride (elements), work items = ceil(M/32) ceil(K/16) 64 }; blk = { job, first block of 256 work items }.
// A Stage-I training step re-packs the forward and the transposed operand of every Bayesian 1x1 layer (the weights are new draws each
// iteration): ~120 launches of a few KiB otherwise.
struct packjob { const float* src; int64_t out; int32_t M, K; int64_t rs, cs, items, pad0, pad1; };
static_assert(sizeof(packjob) == 8 * 8, "packjob is eight 64-bit words (bem.modules.BayesBank builds it as an int64 table)");

__global__ __launch_bounds__(256) void pack_x6_jobs_kernel(const packjob* __restrict__ jobs, const int32_t* __restrict__ blks, float* __restrict__ arena) {
    const packjob jb = jobs[blks[2 * blockIdx.x]];
    const int64_t i = (int64_t)blks[2 * blockIdx.x + 1] * 256 + threadIdx.x;
    if (i >= jb.items) return;
    pack_x6_item(i, jb.src, reinterpret_cast<u32x4*>(arena + jb.out), jb.M, jb.K, (jb.M + 31) / 32, (jb.K + 15) / 16, 0, jb.rs, jb.cs);
}

// Bayesian weight sets straight into operand order: w[set][row][k] = mu + log1p(exp(rho)) * eps, eps injected or drawn
// with the sampler's own Philox stream (element index i = set*M*K + row*K + k, as bem_bnn_sample_f32 numbers it), split
// and stored like pack_x6_kernel does -- the natural-order copy (one write + one read per weight and sample) is skipped.
__device__ __forceinline__ void sample_pack_x6_item(int64_t i, const float* __restrict__ mu, const float* __restrict__ rho,
                                                    const float* __restrict__ eps_in, u32x4* __restrict__ Wp, int M, int K, int MT, int KB,
                                                    uint64_t seed, uint64_t stream_id, int sigma_given) {
    const int lane = (int)(i & 63);
    const int64_t blk = i >> 6;
    const int kb = (int)(blk % KB), mt = (int)((blk / KB) % MT);
    const int64_t set = blk / ((int64_t)KB * MT);
    const int row = mt * 32 + (lane & 31), k0 = kb * 16 + (lane >> 5) * 8;
    float v[8];
    // the lane's 8 consecutive k of one row are 8 consecutive element indices: at most 3 Philox counter blocks
    const int64_t g0 = set * M * K + (int64_t)row * K + k0;
    float z[3][4];
    if (!eps_in && row < M) {
#pragma unroll
        for (int q = 0; q < 3; ++q) philox_normal4((g0 >> 2) + q, seed, stream_id, z[q]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = 0.f;
        if (row < M && k0 + e < K) {
            const int64_t idx = (int64_t)row * K + k0 + e, gi = g0 + e;
            const int off = (int)(g0 & 3);                                // position of the first element in its block
            // z[(off + e) >> 2][(off + e) & 3] with compile-time indices (a runtime index would put z into scratch)
            const float zsel = off == 0 ? z[e >> 2][e & 3] : off == 1 ? z[(e + 1) >> 2][(e + 1) & 3]
                             : off == 2 ? z[(e + 2) >> 2][(e + 2) & 3] : z[(e + 3) >> 2][(e + 3) & 3];
            const float eps = eps_in ? eps_in[gi] : zsel;
            // sigma = log1p(exp(rho)) does not depend on the sample: callers that draw many sets pass it precomputed
            v[e] = mu[idx] + (sigma_given ? rho[idx] : log1pf(expf(rho[idx]))) * eps;
        }
    }
    u32x4 h, m, l;
    split8(v, h, m, l);
    u32x4* o = Wp + ((set * MT + mt) * KB + kb) * 3 * 64 + lane;
    o[0] = h; o[64] = m; o[128] = l;
}

__global__ void sample_pack_x6_kernel(const float* __restrict__ mu, const float* __restrict__ rho, const float* __restrict__ eps_in,
                                      u32x4* __restrict__ Wp, int M, int K, int MT, int KB, int64_t total, uint64_t seed, uint64_t stream_id,
                                      const uint64_t* __restrict__ stream_add, int sigma_given) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (stream_add) stream_id += stream_add[0];                          // device-resident part of the id (see randn_kernel)
    sample_pack_x6_item(i, mu, rho, eps_in, Wp, M, K, MT, KB, seed, stream_id, sigma_given);
}

// ------------------------------------------------------------------------------------------------------------------------
// All Bayesian tensors of a net drawn for a stochastic (eval) forward in ONE launch: the Stage-I net of the Monte-Carlo loop has 60
// Bayesian leaves / 90 tensors, i.e. 90 sampling launches per forward that sit between the layers' own kernels on planes of H/16 x W/16
// pixels, where every dependent launch costs >= 5 us whatever it does.  Segments of a flat output arena:
//   seg = { mu, sig (sigma = log1p(exp(rho)) precomputed for packed 1x1 weights, rho otherwise), first float of the tensor's nsets outputs
//           in the arena, n = elements per set, M, K (packed x6 operand order for the GEMM kernels; K = 0: natural order, as
//           bem_bnn_sample_f32 writes depthwise weights and biases), stream counter, work items, nsets * n }
//   blk = { segment, first work item }: one workgroup = 256 work items of one segment (packed: sample_pack_x6_item; natural: 4 elements)
// Values are those of bem_bnn_sample_pack_x6 (sigma_given) / bem_bnn_sample_f32 for (seed, stream_base + counter).
// ------------------------------------------------------------------------------------------------------------------------
struct ebank_seg { const float* mu; const float* sig; int64_t out; int64_t n; int32_t M, K; uint64_t counter; int64_t items; int64_t total; };
static_assert(sizeof(ebank_seg) == 8 * 8, "ebank_seg is eight 64-bit words (bem.modules.EvalSampleBank builds it as an int64 table)");
struct ebank_blk { int32_t seg; int32_t first; };

__global__ __launch_bounds__(256) void ebank_sample_kernel(const ebank_seg* __restrict__ segs, const ebank_blk* __restrict__ blks,
                                                           float* __restrict__ arena, uint64_t seed, uint64_t stream_base) {
    const ebank_blk bk = blks[blockIdx.x];
    const ebank_seg sg = segs[bk.seg];
    const int64_t i = (int64_t)bk.first * 256 + threadIdx.x;
    if (i >= sg.items) return;
    const uint64_t sid = stream_base + sg.counter;
    float* out = arena + sg.out;
    if (sg.K > 0) {
        sample_pack_x6_item(i, sg.mu, sg.sig, nullptr, reinterpret_cast<u32x4*>(out), sg.M, sg.K, (sg.M + 31) / 32, (sg.K + 15) / 16, seed, sid, 1);
        return;
    }
    const int64_t i0 = 4 * i, total = sg.total;                                       // natural order: total = nsets * n elements
    float z[4];
    philox_normal4(i, seed, sid, z);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t e = i0 + q;
        if (e < total) {
            const int64_t k = e % sg.n;
            out[e] = sg.mu[k] + log1pf(expf(sg.sig[k])) * z[q];
        }
    }
}

}  // namespace

extern "C" int bem_bnn_ebank_sample_f32(const void* segs, const void* blks, int nblk, float* arena, uint64_t seed, uint64_t stream_base,
                                        void* stream) {
    BEM_REQUIRE(segs && blks && arena && nblk > 0, "bnn_ebank_sample: bad arguments");
    BEM_REQUIRE(((uintptr_t)arena & 15) == 0, "bnn_ebank_sample: the arena must be 16-byte aligned");
    ebank_sample_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>((const ebank_seg*)segs, (const ebank_blk*)blks, arena, seed, stream_base);
    return bem_check_launch("bnn_ebank_sample");
}

extern "C" int bem_bnn_sample_pack_x6(const float* mu, const float* rho, const float* eps_in, float* Wp, int nsets, int M, int K,
                                      uint64_t seed, uint64_t stream_id, const uint64_t* stream_add, int sigma_given, void* stream) {
    BEM_REQUIRE(mu && rho && Wp, "bnn_sample_pack_x6: null tensor");
    BEM_REQUIRE(nsets >= 0 && M > 0 && K > 0, "bnn_sample_pack_x6: bad shape");
    BEM_REQUIRE(((uintptr_t)Wp & 15) == 0, "bnn_sample_pack_x6: output must be 16-byte aligned");
    if (nsets == 0) return BEM_OK;
    const int MT = cdiv(M, 32), KB = cdiv(K, 16);
    const int64_t total = (int64_t)nsets * MT * KB * 64;
    sample_pack_x6_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, (hipStream_t)stream>>>(mu, rho, eps_in, reinterpret_cast<u32x4*>(Wp), M, K, MT, KB,
                                                                                       total, seed, stream_id, stream_add, sigma_given);
    return bem_check_launch("bnn_sample_pack_x6");
}

extern "C" int64_t bem_pw_x6_packed_elems(int M, int K) {       // in floats (4 per 16-byte vector)
    return (int64_t)cdiv(M, 32) * cdiv(K, 16) * 3 * 64 * 4;
}

extern "C" int bem_pack_pw_weight_x6_strided(const float* W, float* Wp, int nsets, int M, int K, int64_t set_stride, int64_t row_stride,
                                             int64_t col_stride, void* stream) {
    BEM_REQUIRE(W && Wp, "pack_pw_weight_x6: null tensor");
    BEM_REQUIRE(nsets >= 0 && M > 0 && K > 0 && set_stride >= 0 && row_stride >= 0 && col_stride >= 0, "pack_pw_weight_x6: bad shape / strides");
    BEM_REQUIRE(((uintptr_t)Wp & 15) == 0, "pack_pw_weight_x6: output must be 16-byte aligned");
    if (nsets == 0) return BEM_OK;
    const int MT = cdiv(M, 32), KB = cdiv(K, 16);
    const int64_t total = (int64_t)nsets * MT * KB * 64;
    pack_x6_kernel<<<(unsigned)cdiv64(total, 256), 256, 0, (hipStream_t)stream>>>(W, reinterpret_cast<u32x4*>(Wp), M, K, MT, KB, total, set_stride,
                                                                                 row_stride, col_stride);
    return bem_check_launch("pack_pw_weight_x6");
}

extern "C" int bem_pack_pw_weight_x6_jobs(const void* jobs, const void* blks, int nblk, float* arena, void* stream) {
    BEM_REQUIRE(jobs && blks && arena && nblk > 0, "pack_pw_weight_x6_jobs: bad arguments");
    BEM_REQUIRE(((uintptr_t)arena & 15) == 0, "pack_pw_weight_x6_jobs: the arena must be 16-byte aligned");
    pack_x6_jobs_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>((const packjob*)jobs, (const int32_t*)blks, arena);
    return bem_check_launch("pack_pw_weight_x6_jobs");
}

extern "C" int bem_pack_pw_weight_x6(const float* W, float* Wp, int nsets, int M, int K, void* stream) {
    return bem_pack_pw_weight_x6_strided(W, Wp, nsets, M, K, (int64_t)M * K, K, 1, stream);
}

extern "C" int bem_pw_gemm_x6_f32(const bem_pw_args* a, void* stream) {
    BEM_REQUIRE(a, "pw_gemm_x6: null args");
    BEM_REQUIRE(a->x1 && a->Wp && a->out, "pw_gemm_x6: null tensor");
    BEM_REQUIRE(a->B >= 0 && a->B <= 65535 && a->M > 0 && a->M <= BEM_X6_MAXM && a->K > 0 && a->L >= 0, "pw_gemm_x6: bad shape B=%d M=%d K=%d L=%d", a->B, a->M, a->K, a->L);
    BEM_REQUIRE(a->in_mode >= 0 && a->in_mode <= 2, "pw_gemm_x6: in_mode %d", a->in_mode);
    if (a->in_mode == 0) BEM_REQUIRE(a->K == a->C1, "pw_gemm_x6: K %d != C1 %d", a->K, a->C1);
    if (a->in_mode == 1) BEM_REQUIRE(a->x2 && a->K == a->C1 && a->C1 == a->C2, "pw_gemm_x6: sum mode needs x2 and K == C1 == C2");
    if (a->in_mode == 2) BEM_REQUIRE(a->x2 && a->K == a->C1 + a->C2, "pw_gemm_x6: cat mode needs x2 and K == C1 + C2");
    BEM_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr), "pw_gemm_x6: ln_w / ln_b must both be set or both NULL");
    BEM_REQUIRE(a->act == 0 || (a->act == 1 && a->prelu), "pw_gemm_x6: act %d", a->act);
    BEM_REQUIRE(a->out_mode == 0 || (a->out_mode == 1 && a->M % 4 == 0 && a->Win > 0 && a->L % a->Win == 0 && !a->res),
                "pw_gemm_x6: out_mode %d constraints", a->out_mode);
    BEM_REQUIRE(((uintptr_t)a->Wp & 15) == 0 && a->w_bstride % 4 == 0, "pw_gemm_x6: packed weights must be 16-byte aligned");
    BEM_REQUIRE((int64_t)a->M * a->L < (1ll << 30), "pw_gemm_x6: M * L = %lld exceeds the 32-bit lane offsets of the epilogue", (long long)a->M * a->L);
    const bool ln = a->ln_w != nullptr;
    BEM_REQUIRE(!ln || a->K <= BEM_X6_MAXK_LN, "pw_gemm_x6: LayerNorm prologue supports K <= %d (got %d)", BEM_X6_MAXK_LN, a->K);
    if (a->B == 0 || a->L == 0) return BEM_OK;
    PwX k;
    k.x1 = a->x1; k.x2 = a->x2 ? a->x2 : a->x1; k.C1 = a->C1; k.C2 = a->x2 ? a->C2 : a->C1; k.in_mode = a->in_mode;
    k.ln_w = a->ln_w; k.ln_b = a->ln_b; k.ln_eps = a->ln_eps;
    k.Wp = reinterpret_cast<const u32x4*>(a->Wp); k.w_bstride = a->w_bstride / 4; k.bias = a->bias; k.bias_bstride = a->bias_bstride;
    k.res = a->res; k.prelu = a->prelu; k.act = a->act; k.out = a->out; k.out_mode = a->out_mode; k.Win = a->Win;
    k.M = a->M; k.K = a->K; k.L = a->L; k.KB = cdiv(a->K, 16); k.MT = cdiv(a->M, 32); k.mtpb = k.MT;
    hipStream_t s = (hipStream_t)stream;
    const bool sum = a->in_mode == 1;
    const bool al = (((uintptr_t)a->x1 | (uintptr_t)(a->x2 ? a->x2 : a->x1) | (uintptr_t)a->out | (uintptr_t)(a->res ? a->res : a->out)) & 7) == 0;
    const bool vec = a->L % 2 == 0 && al;
#define BEM_X6_RES(KBM, NSUB, MTW)                                                                       \
    do {                                                                                                 \
        dim3 grid(cdiv(a->L, 128 * NSUB), 1, a->B);                                                      \
        /* few pixels (Stage I: 4x4 ... 16x16 maps), many output rows: slice M over grid.y so that the weight stream of  \
           one sample is pulled by several workgroups (each repeats the cheap LayerNorm of the same pixels) */            \
        const int64_t wv = (int64_t)a->B * cdiv(a->L, 32 * NSUB);                                        \
        const int groups = cdiv(k.MT, MTW);                                                              \
        const int ny = wv >= 2048 ? 1 : (int)std::min<int64_t>(groups, cdiv64(2048, wv));                \
        k.mtpb = cdiv(groups, ny) * MTW;                                                                 \
        grid.y = cdiv(k.MT, k.mtpb);                                                                     \
        if (NSUB == 2 && vec && !sum) pw_x6_res_kernel<KBM, NSUB, MTW, false, true><<<grid, 256, 0, s>>>(k);   \
        else if (NSUB == 2 && vec) pw_x6_res_kernel<KBM, NSUB, MTW, true, true><<<grid, 256, 0, s>>>(k);       \
        else if (!sum) pw_x6_res_kernel<KBM, NSUB, MTW, false, false><<<grid, 256, 0, s>>>(k);                 \
        else pw_x6_res_kernel<KBM, NSUB, MTW, true, false><<<grid, 256, 0, s>>>(k);                            \
        return bem_check_launch("pw_x6_res");                                                            \
    } while (0)
    // one M-tile at a time where a wave holds two sub-tiles: the accumulator pairs double the register cost of an M-tile
    if (k.KB <= 3) BEM_X6_RES(3, 2, 1);
    if (ln && k.KB <= 5) BEM_X6_RES(5, 2, 1);
    static const bool res10 = !(getenv("BEM_X6_RES10") && atoi(getenv("BEM_X6_RES10")) == 0);       // A/B: the two-sweep streaming form instead
    static const bool reslds = !(getenv("BEM_X6_RES_LDS") && atoi(getenv("BEM_X6_RES_LDS")) == 0);   // A/B: every wave streams its own weights from L2
    if (ln && k.KB == 10 && k.K == 160 && k.MT >= 8 && res10 && reslds && (int64_t)a->B * cdiv(a->L, 32) >= 2048) {
        // many M-tiles over a full-width K and enough pixels that a workgroup walks all of them: an M-tile's weights go through LDS once per
        // workgroup (pw_x6_res_lds_kernel; level-2 project_in 335 -> 188 us).  With few pixels (Stage I) M is sliced over grid.y and the
        // per-wave streaming form below is the faster one (50 vs 63 us).
        dim3 grid(cdiv(a->L, 128), 1, a->B);
        k.mtpb = k.MT;
        if (!sum) pw_x6_res_lds_kernel<10, false, false><<<grid, 256, 0, s>>>(k);       // 32 pixels per wave: the scalar-pixel forms, as in BEM_X6_RES(10, 1, *)
        else pw_x6_res_lds_kernel<10, true, false><<<grid, 256, 0, s>>>(k);
        return bem_check_launch("pw_x6_res_lds");
    }
    if (ln && k.KB <= 10 && res10) { if (k.MT == 1) BEM_X6_RES(10, 1, 1); else BEM_X6_RES(10, 1, 2); }
#undef BEM_X6_RES
    {
        // M-tiles per pass over x: every extra grid.y slice re-reads the input.  Two with 64-pixel waves; for exactly three
        // M-tiles (level-1 project_out, M = 80) three with 32-pixel waves -- half the bytes per load / store instruction,
        // but x is read once instead of twice
        // (measured: 227 us vs 219 us for the two-slice form at K = 320, M = 80, 64x64 -- the re-read is served by L2; kept off)
        const bool three = false && k.MT == 3 && !ln;
        const int mtw = three ? 3 : (k.MT == 1 ? 1 : 2);
        dim3 grid(cdiv(a->L, three ? 128 : 256), cdiv(k.MT, mtw), a->B);
#define BEM_X6_STREAM(MTW, LN)                                                                \
    do {                                                                                      \
        if (vec && !sum) pw_x6_stream_kernel<MTW, 2, false, true, LN><<<grid, 256, 0, s>>>(k);   \
        else if (vec) pw_x6_stream_kernel<MTW, 2, true, true, LN><<<grid, 256, 0, s>>>(k);       \
        else if (!sum) pw_x6_stream_kernel<MTW, 2, false, false, LN><<<grid, 256, 0, s>>>(k);    \
        else pw_x6_stream_kernel<MTW, 2, true, false, LN><<<grid, 256, 0, s>>>(k);               \
    } while (0)
        if (three) {
            if (!sum) pw_x6_stream_kernel<3, 1, false, false, false><<<grid, 256, 0, s>>>(k);
            else pw_x6_stream_kernel<3, 1, true, false, false><<<grid, 256, 0, s>>>(k);
        }
        else if (ln) { if (mtw == 1) BEM_X6_STREAM(1, true); else BEM_X6_STREAM(2, true); }
        else { if (mtw == 1) BEM_X6_STREAM(1, false); else BEM_X6_STREAM(2, false); }
#undef BEM_X6_STREAM
    }
    return bem_check_launch("pw_x6_stream");
}

// ================================================================================================
// Dense convolutions as shifted 1x1 GEMM taps on the same x6 machinery (3x3 stride 1 pad 1; 4x4 stride 2 pad 1):
//     out[co][p] = relu?( sum_{tap} sum_ci W[co][ci][tap] * x[ci][S*p + tap offset] + bias[co] ) + res1 + res2
// No im2col patch: tap (ky, kx) reads the input pixels of the wave's 64 output pixels straight from global memory (the
// displaced reads of a k-block overlap and are served by L1 / L2), masks the pixels that fall outside the image, splits
// them into bf16 limbs and issues the six limb products against that tap's weight block.  The x loads of the next tap
// are requested before the MFMAs of the current one.  Wp: (KH*KW taps, MT, KB, 3 limbs, 64 lanes) 16-byte vectors =
// bem_pack_pw_weight_x6 of the (KH*KW, Cout, Cin) tap matrices.  Requires an even output width and Cin % 8 == 0.
// ================================================================================================
namespace {

struct CvX {
    const float* x; int64_t x_bs;
    const u32x4* Wp; const float* bias; const float* res1; const float* res2; float* out;
    int Cin, H, W, Ho, Wo, Cout, KB, MT, relu, pad, dil;
};

template <int MTW, int KH, int KW, int S>
__global__ __launch_bounds__(256, 2) void conv_taps_x6_kernel(CvX k) {
    constexpr int NSUB = 2, NTAP = KH * KW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z, mt0 = blockIdx.y * MTW;
    const int Lo = k.Ho * k.Wo, Li = k.H * k.W;
    const int p0 = (xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * 64;
    if (p0 >= Lo) return;
    const int p = p0 + 2 * n;                       // this lane's two output pixels p, p + 1 (same row: Wo is even)
    const bool live = p < Lo;
    const int pc = live ? p : 0;
    const int yo = pc / k.Wo, xo = pc - yo * k.Wo;
    const int yi0 = yo * S - k.pad, xi0 = xo * S - k.pad;      // input position of tap (0, 0) for the first pixel
    const float* xb = k.x + (int64_t)b * k.x_bs;
    f32x16 acc[MTW][NSUB], alo[MTW][NSUB];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NSUB; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = alo[m][t][r] = 0.f;
    const u32x4* wbase = k.Wp + lane;
    const int64_t tap_stride = (int64_t)k.MT * k.KB * 3 * 64, mt_stride = (int64_t)k.KB * 3 * 64;
    const int nsteps = k.KB * NTAP;
    // step s = kb * NTAP + tap.  Loads of a step: 8 channels (16 kb + 8 kh + e) at the two input pixels of this tap.
    auto load_x = [&](int s, float (&dst)[8][NSUB], float (&mk)[NSUB]) {
        const int kb = min(s / NTAP, k.KB - 1), tap = s - (s / NTAP) * NTAP;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int yy = yi0 + ky * k.dil, x0 = xi0 + kx * k.dil, x1 = x0 + S;
        const bool rowok = live && yy >= 0 && yy < k.H;
        mk[0] = (rowok && x0 >= 0 && x0 < k.W) ? 1.f : 0.f;
        mk[1] = (rowok && x1 >= 0 && x1 < k.W) ? 1.f : 0.f;
        const int q = yy * k.W + x0;
        // the upper half-wave reads channels + 8; past Cin (a half-filled last k-block) it re-reads the lower half,
        // whose weights there are zero
        const int c0 = 16 * kb, hoff = (c0 + 8 < k.Cin) ? 8 * kh : 0;
        if (S == 1) {
            // the pair (q, q + 1) as ONE 8-byte load (global loads need dword alignment only) at a base clamped into the
            // plane; d = q - base is 0 except at the two ends of the plane, where one of the two pixels is outside anyway
            const int qb = min(max(q, 0), Li - 2), d = q - qb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* pl = xb + (int64_t)(min(c0 + e, k.Cin - 1 - hoff) + hoff) * Li;
                float2 v;
                __builtin_memcpy(&v, pl + qb, sizeof(v));
                dst[e][0] = d > 0 ? v.y : v.x;
                dst[e][1] = d < 0 ? v.x : v.y;
            }
        } else {
            const int q0 = min(max(q, 0), Li - 1), q1 = min(max(q + S, 0), Li - 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float* pl = xb + (int64_t)(min(c0 + e, k.Cin - 1 - hoff) + hoff) * Li;
                dst[e][0] = pl[q0];
                dst[e][1] = pl[q1];
            }
        }
    };
    auto load_w = [&](int s, u32x4 (&dst)[MTW][3]) {
        const int sc = min(s, nsteps - 1), kb = sc / NTAP, tap = sc - kb * NTAP;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const bool ok = s < nsteps && mt0 + m < k.MT;
            const u32x4* wp = wbase + tap * tap_stride + (int64_t)(mt0 + m < k.MT ? mt0 + m : 0) * mt_stride + (int64_t)kb * 3 * 64;
            const uint32_t mk = ok ? 0xffffffffu : 0u;
#pragma unroll
            for (int li = 0; li < 3; ++li) {
                const u32x4 w = wp[li * 64];
                dst[m][li] = u32x4{w[0] & mk, w[1] & mk, w[2] & mk, w[3] & mk};
            }
        }
    };
    float xn[8][NSUB], mkn[NSUB];
    u32x4 wn[MTW][3];
    load_x(0, xn, mkn);
    load_w(0, wn);
    for (int s = 0; s < nsteps; ++s) {
        u32x4 xl[NSUB][3], wc[MTW][3];
#pragma unroll
        for (int t = 0; t < NSUB; ++t) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = xn[e][t] * mkn[t];
            split8(v, xl[t][0], xl[t][1], xl[t][2]);
        }
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int li = 0; li < 3; ++li) wc[m][li] = wn[m][li];
        load_x(s + 1, xn, mkn);          // past the end: clamped, never used
        load_w(s + 1, wn);
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int t = 0; t < NSUB; ++t) mac6(wc[m], xl[t], acc[m][t], alo[m][t]);
    }
    // epilogue: out = relu?(acc + bias) + res1 + res2, plane bases uniform, one lane offset
    const float lo = k.relu ? 0.f : -3.402823466e38f;
    const uint32_t loff = (uint32_t)(4 * kh) * (uint32_t)Lo + (uint32_t)pc;
    float* outb = k.out + (int64_t)b * k.Cout * Lo;
    const float* r1b = k.res1 ? k.res1 + (int64_t)b * k.Cout * Lo : nullptr;
    const float* r2b = k.res2 ? k.res2 + (int64_t)b * k.Cout * Lo : nullptr;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        if (mt0 + m >= k.MT) continue;
        const int rb = (mt0 + m) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float bv[4], rv[4][NSUB];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int urow = rb + 8 * g + i;
                const int lrow = min(urow + 4 * kh, k.Cout - 1);
                bv[i] = k.bias ? k.bias[lrow] : 0.f;
                rv[i][0] = rv[i][1] = 0.f;
                const uint32_t ro = (uint32_t)lrow * (uint32_t)Lo + (uint32_t)pc;
                if (r1b) { const float2 q = *reinterpret_cast<const float2*>(r1b + ro); rv[i][0] += q.x; rv[i][1] += q.y; }
                if (r2b) { const float2 q = *reinterpret_cast<const float2*>(r2b + ro); rv[i][0] += q.x; rv[i][1] += q.y; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * g + i, urow = rb + 8 * g + i;
                const float o0 = fmaxf(acc[m][0][r] + alo[m][0][r] + bv[i], lo) + rv[i][0];
                const float o1 = fmaxf(acc[m][1][r] + alo[m][1][r] + bv[i], lo) + rv[i][1];
                if (live && urow + 4 * kh < k.Cout) *reinterpret_cast<float2*>(outb + (int64_t)urow * Lo + loff) = make_float2(o0, o1);
            }
        }
    }
}

}  // namespace

static int conv_taps_launch(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1, const float* res2,
                            float* out, int B, int Cin, int H, int W, int Cout, int KH, int stride, int dil, int relu, void* stream, const char* what) {
    BEM_REQUIRE(x && Wp && out, "%s: null tensor", what);
    BEM_REQUIRE(B >= 0 && B <= 65535 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "%s: bad shape", what);
    BEM_REQUIRE((KH == 3 && stride == 1 && (dil == 1 || dil == 2)) || (KH == 3 && stride == 2 && dil == 1),
                "%s: supported forms are 3x3 s1 (dilation 1 / 2, padding = dilation) and 3x3 s2 p1", what);
    const int pad = dil;                                        // "same" padding of the dilated 3x3; 1 for the others
    const int Ho = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (KH - 1) - 1) / stride + 1;
    BEM_REQUIRE(Ho > 0 && Wo > 0 && Wo % 2 == 0 && Cin % 8 == 0 && H * W >= 2, "%s: needs an even output width and Cin %% 8 == 0 (got Wo=%d Cin=%d)", what, Wo, Cin);
    BEM_REQUIRE(((uintptr_t)Wp & 15) == 0 && (((uintptr_t)out | (uintptr_t)(res1 ? res1 : out) | (uintptr_t)(res2 ? res2 : out)) & 7) == 0,
                "%s: alignment (packed weights 16 bytes, out / residuals 8 bytes)", what);
    BEM_REQUIRE((int64_t)Cout * Ho * Wo < (1ll << 30) && (int64_t)Cin * H * W < (1ll << 30), "%s: plane set too large for 32-bit lane offsets", what);
    if (B == 0) return BEM_OK;
    CvX k;
    k.x = x; k.x_bs = x_bstride; k.Wp = reinterpret_cast<const u32x4*>(Wp); k.bias = bias; k.res1 = res1; k.res2 = res2; k.out = out;
    k.Cin = Cin; k.H = H; k.W = W; k.Ho = Ho; k.Wo = Wo; k.Cout = Cout; k.KB = cdiv(Cin, 16); k.MT = cdiv(Cout, 32); k.relu = relu; k.pad = pad; k.dil = dil;
    const int mtw = k.MT == 1 ? 1 : 2;
    dim3 grid(cdiv(Ho * Wo, 256), cdiv(k.MT, mtw), B);
    hipStream_t s = (hipStream_t)stream;
    if (KH == 3 && stride == 1) {
        if (mtw == 1) conv_taps_x6_kernel<1, 3, 3, 1><<<grid, 256, 0, s>>>(k);
        else conv_taps_x6_kernel<2, 3, 3, 1><<<grid, 256, 0, s>>>(k);
    } else {
        if (mtw == 1) conv_taps_x6_kernel<1, 3, 3, 2><<<grid, 256, 0, s>>>(k);
        else conv_taps_x6_kernel<2, 3, 3, 2><<<grid, 256, 0, s>>>(k);
    }
    return bem_check_launch(what);
}

extern "C" int bem_conv4x4s2_fast_supported(int Cin, int H, int W);
extern "C" int bem_conv3x3_rows_supported(int Cin, int H, int W);
int conv_rows_launch(int KS, const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1, const float* res2, float* out,
                     int B, int Cin, int H, int W, int Cout, int relu, void* stream);                    // conv_rows_x6.hip
static bool rows_aligned(const float* x, int64_t x_bstride, const float* out, const float* res1, const float* res2) {
    return x && out && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)(res1 ? res1 : out) | (uintptr_t)(res2 ? res2 : out)) & 15) == 0 && x_bstride % 4 == 0;
}

extern "C" int bem_conv3x3_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1,
                                  const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int relu, void* stream) {
    // the row form (conv_rows_x6.hip) where the shape allows; nine shifted taps otherwise (BEM_CONV3_ROWS=0: always)
    static const bool rows = !(getenv("BEM_CONV3_ROWS") && atoi(getenv("BEM_CONV3_ROWS")) == 0);
    if (rows && Cin > 0 && bem_conv3x3_rows_supported(Cin, H, W) && rows_aligned(x, x_bstride, out, res1, res2))
        return conv_rows_launch(3, x, x_bstride, Wp, bias, res1, res2, out, B, Cin, H, W, Cout, relu, stream);
    return conv_taps_launch(x, x_bstride, Wp, bias, res1, res2, out, B, Cin, H, W, Cout, 3, 1, 1, relu, stream, "conv3x3_x6");
}


extern "C" int bem_conv4x4s2_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1,
                                    const float* res2, float* out, int B, int Cin, int H, int W, int Cout, int relu, void* stream) {
    // the row form (conv_rows_x6.hip); shapes outside it (bem_conv4x4s2_fast_supported == 0) belong to bem_conv2d_mfma_f32
    BEM_REQUIRE(Cin > 0 && bem_conv4x4s2_fast_supported(Cin, H, W) && rows_aligned(x, x_bstride, out, res1, res2),
                "conv4x4s2_x6: needs W = 2 Wo with Wo a power of two <= 64, even H, Cin %% 8 == 0 and 16-byte aligned tensors");
    return conv_rows_launch(4, x, x_bstride, Wp, bias, res1, res2, out, B, Cin, H, W, Cout, relu, stream);
}

extern "C" int bem_conv_taps_x6_f32(const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1, const float* res2,
                                    float* out, int B, int Cin, int H, int W, int Cout, int K, int stride, int dilation, int relu, void* stream) {
    return conv_taps_launch(x, x_bstride, Wp, bias, res1, res2, out, B, Cin, H, W, Cout, K, stride, dilation, relu, stream, "conv_taps_x6");
}
