// Fused gated-dconv MLP (gdMlp, basicsr/vmamba/models/vmamba.py:116-133) with its LayerNorm prologue and
// residual epilogue:
//     out = x + W_o * ( GELU(dw3x3(h)[0:Hd]) * dw3x3(h)[Hd:2Hd] ) + b_o ,   h = W_i * LN(x) + b_i
// The 2*Hd-channel intermediate h (8C channels, the widest tensor of the network) never leaves the CU:
// per 16 gate channels a (16 + 16)-row slab of h is produced by the f32 matrix cores into LDS for one
// spatial tile plus a one-pixel halo, the depthwise 3x3 + gate runs on it from LDS, and the result feeds the
// matrix cores again as a 16-deep slice of the W_o contraction, accumulated in registers.
//
//   phase 0 (once)   LN(x) of the (TH+2) x (TW+2) halo tile -> LDS  xn[C][HPp]
//   per chunk ch of 16 gate channels:
//     P1  MFMA   hh[32][HPp] = Wi'[ch] (32 x C) * xn (+ b_i, zero outside the image = conv zero padding)
//     P2  VALU   gg[16][TP]  = GELU(dw(hh[0:16])) * dw(hh[16:32])       (TP = TH*TW interior pixels)
//     P3  MFMA   acc[C][TP] += W_o[:, 16ch:16ch+16] * gg
//   epilogue         out = acc + b_o + x
// Wi' is W_i with rows regrouped so that M-tile ch = [a-rows 16ch..16ch+15 | b-rows Hd+16ch..Hd+16ch+15]
// (bem_pack_pw_weight_gate_f32).  Algorithmic HBM traffic: read x once (+ halo), write out once.
#include "bem_common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GdK {
    const float* x; float* out;
    const float* ln_w; const float* ln_b; float eps;
    const float* Wpi; int64_t wpi_bs; const float* bpi; int64_t bpi_bs;
    const float* dww; int64_t dww_bs; const float* dwb; int64_t dwb_bs;
    const float* Wpo; int64_t wpo_bs; const float* bpo; int64_t bpo_bs;
    int C, Hd, H, W, KS, KSo, MTo, tilesX;
    unsigned long long* stamps;   // debug: per-phase s_memtime stamps of one workgroup (BEM_GD_DBG bit 4)
    int dbg;   // timing experiments only (BEM_GD_DBG): bit0 skip P1 MFMA, bit1 skip P2, bit2 skip P3, bit3 skip hh write
};

constexpr int TW = 16, HW2 = TW + 2;
constexpr int MTO_LIMIT = 5;   // C <= 160

#define GD_STAMP(i)                                                                                      \
    do {                                                                                                 \
        if (k.stamps && blockIdx.x == 37 && blockIdx.y == 3 && threadIdx.x == 0 && (i) < 96)              \
            k.stamps[(i)] = __builtin_amdgcn_s_memtime();                                                \
    } while (0)

template <int TH, int MTO_MAX>
__global__ __launch_bounds__(256, 2) void gdmlp_fused_kernel(GdK k) {
    constexpr int HP = (TH + 2) * HW2;
    constexpr int HPp = ((HP + 31) / 32) * 32;
    constexpr int NT1 = HPp / 32;
    constexpr int TP = TH * TW;
    constexpr int NT3 = TP / 32;
    constexpr int N1W = (NT1 + 3) / 4;          // halo N-tiles per wave (max)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Kp = 2 * k.KS;
    float* xn = sm;                              // [Kp][HPp]
    float* hh = xn + (size_t)Kp * HPp;           // [32][HPp]
    float* gg = hh + 32 * HPp;                   // [16][TP]
    float* st = gg + 16 * TP;                    // mean[HPp], rstd[HPp]
    float* lnp = st + 2 * HPp;                   // ln_w[Kp], ln_b[Kp]
    float* bsm = lnp + 2 * Kp;                   // project_in bias [2*Hd]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / k.tilesX) * TH, tx0 = (blockIdx.x % k.tilesX) * TW;
    const int64_t HWp = (int64_t)k.H * k.W;
    const float* xb = k.x + (int64_t)b * k.C * HWp;

    GD_STAMP(0);
    // ---------------- phase 0: halo tile of x -> LDS, LayerNorm in place ----------------
    {
        // one (channel, halo row) per work item: 16 aligned interior floats as 4 float4 + the two halo columns;
        // every load of a batch is issued before the first LDS store (one memory latency per batch of 512 rows).
        constexpr int ROWS = TH + 2;
        constexpr int MAXIT = 2;
        const int nrows = k.C * ROWS;
        const bool fast = (k.W % 4 == 0);
        for (int base = 0; base < nrows; base += 256 * MAXIT) {
            float4 v4[MAXIT][4];
            float vl[MAXIT], vr[MAXIT];
#pragma unroll
            for (int u = 0; u < MAXIT; ++u) {
                const int it = base + u * 256 + threadIdx.x;
                const int c = it / ROWS, hy = it - c * ROWS;
                const int iy = ty0 - 1 + hy;
                const bool rowok = it < nrows && iy >= 0 && iy < k.H;
                const float* rp = xb + (int64_t)(rowok ? c : 0) * HWp + (int64_t)(rowok ? iy : 0) * k.W;
                vl[u] = (rowok && tx0 > 0) ? rp[tx0 - 1] : 0.f;
                vr[u] = (rowok && tx0 + TW < k.W) ? rp[tx0 + TW] : 0.f;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const int ix = tx0 + 4 * f;
                    if (rowok && fast && ix + 3 < k.W) {
                        v4[u][f] = *reinterpret_cast<const float4*>(rp + ix);
                    } else {
                        v4[u][f].x = (rowok && ix < k.W) ? rp[ix] : 0.f;
                        v4[u][f].y = (rowok && ix + 1 < k.W) ? rp[ix + 1] : 0.f;
                        v4[u][f].z = (rowok && ix + 2 < k.W) ? rp[ix + 2] : 0.f;
                        v4[u][f].w = (rowok && ix + 3 < k.W) ? rp[ix + 3] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < MAXIT; ++u) {
                const int it = base + u * 256 + threadIdx.x;
                if (it < nrows) {
                    const int c = it / ROWS, hy = it - c * ROWS;
                    float* d = xn + c * HPp + hy * HW2;
                    d[0] = vl[u];
#pragma unroll
                    for (int f = 0; f < 4; ++f) { d[1 + 4 * f] = v4[u][f].x; d[2 + 4 * f] = v4[u][f].y; d[3 + 4 * f] = v4[u][f].z; d[4 + 4 * f] = v4[u][f].w; }
                    d[HW2 - 1] = vr[u];
                }
            }
        }
        // zero the pad columns [HP, HPp) of every channel row and the pad channel rows [C, Kp)
        for (int idx = threadIdx.x; idx < Kp * (HPp - HP); idx += 256) {
            const int c = idx / (HPp - HP), q = HP + idx - c * (HPp - HP);
            xn[c * HPp + q] = 0.f;
        }
        for (int idx = k.C * HPp + threadIdx.x; idx < Kp * HPp; idx += 256) xn[idx] = 0.f;
        for (int c = threadIdx.x; c < Kp; c += 256) {
            lnp[c] = c < k.C ? k.ln_w[c] : 0.f;
            lnp[Kp + c] = c < k.C ? k.ln_b[c] : 0.f;
        }
        {
            const float* bp = k.bpi + (int64_t)b * k.bpi_bs;
            for (int i = threadIdx.x; i < 2 * k.Hd; i += 256) bsm[i] = bp[i];
        }
        __syncthreads();
        GD_STAMP(1);
        for (int q = threadIdx.x; q < HPp; q += 256) {
            float s = 0.f;
            for (int c = 0; c < k.C; ++c) s += xn[c * HPp + q];
            const float mean = s / (float)k.C;
            float var = 0.f;
            for (int c = 0; c < k.C; ++c) {
                const float d = xn[c * HPp + q] - mean;
                var = fmaf(d, d, var);
            }
            st[q] = mean;
            st[HPp + q] = 1.f / sqrtf(var / (float)k.C + k.eps);
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < k.C * HPp; idx += 256) {
            const int c = idx / HPp, q = idx - c * HPp;
            xn[idx] = (xn[idx] - st[q]) * st[HPp + q] * lnp[c] + lnp[Kp + c];
        }
        __syncthreads();
        GD_STAMP(2);
    }

    const float* wpi = k.Wpi + (int64_t)b * k.wpi_bs + lane;
    const float* dww = k.dww + (int64_t)b * k.dww_bs;
    const float* dwb = k.dwb ? k.dwb + (int64_t)b * k.dwb_bs : nullptr;
    const float* wpo = k.Wpo + (int64_t)b * k.wpo_bs + lane;

    f32x16 acc3[MTO_MAX];
#pragma unroll
    for (int m = 0; m < MTO_MAX; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[m][r] = 0.f;

    // which halo pixels of this wave's N-tiles are inside the image (zero padding of the depthwise conv)
    bool inside[N1W];
#pragma unroll
    for (int t = 0; t < N1W; ++t) {
        const int q = 32 * (wave + 4 * t) + j;
        const int hy = q / HW2, hx = q - hy * HW2;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        inside[t] = (wave + 4 * t < NT1) && q < HP && iy >= 0 && iy < k.H && ix >= 0 && ix < k.W;
    }

    const int nchunks = k.Hd / 16;
    constexpr int PB = 4;                          // k-steps per P1 operand batch; two batches are kept in flight
    const int64_t mts = (int64_t)k.KSo * 64;
    const int nb = (k.KS + PB - 1) / PB;           // batches per chunk
    const int total_b = nb * nchunks;              // batch index space over all chunks
    // A operand of P1 for global batch g (chunk g / nb, k-steps (g % nb) * PB ...): zero beyond KS so that the MFMA
    // loop needs no per-step branch (a branch per k-step costs more than the wasted MFMA issue slots).
    auto load_a = [&](int g, float (&dst)[PB]) {
        const int cg = g / nb, s0 = (g - cg * nb) * PB;
#pragma unroll
        for (int u = 0; u < PB; ++u)        // clamped address + multiplicative mask (uniform selects turn into branch + vmcnt(0))
            dst[u] = wpi[((int64_t)min(cg, nchunks - 1) * k.KS + min(s0 + u, k.KS - 1)) * 64] * ((g < total_b && s0 + u < k.KS) ? 1.f : 0.f);
    };
    float a_n0[PB], a_n1[PB];
    load_a(0, a_n0);
    load_a(1, a_n1);

    for (int ch = 0; ch < nchunks; ++ch) {
        // ---------------- P1: hh = Wi'[ch] * xn  (M-tile ch of the regrouped project_in weight) ----------------
        {
            f32x16 acc1[N1W];
#pragma unroll
            for (int t = 0; t < N1W; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
            for (int bi = 0; bi < nb; ++bi) {
                float ac[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) { ac[u] = a_n0[u]; a_n0[u] = a_n1[u]; }
                load_a(ch * nb + bi + 2, a_n1);
                if (!(k.dbg & 1)) {
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        const int krow = min(2 * (bi * PB + u) + half, Kp - 1);     // rows past K pair with a zero A operand
                        const float* xr = xn + krow * HPp + j;
#pragma unroll
                        for (int t = 0; t < N1W; ++t) {
                            // waves whose second tile does not exist recompute their first one into a scratch accumulator
                            const int nt = (wave + 4 * t < NT1) ? (wave + 4 * t) : wave;
                            acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[u], xr[32 * nt], acc1[t], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < N1W; ++t) {
                if (wave + 4 * t < NT1 && !(k.dbg & 8)) {
                    const int q = 32 * (wave + 4 * t) + j;
                    float bv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        bv[r] = bsm[(row < 16) ? (ch * 16 + row) : (k.Hd + ch * 16 + row - 16)];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        hh[row * HPp + q] = inside[t] ? (acc1[t][r] + bv[r]) : 0.f;
                    }
                }
            }
        }
        GD_STAMP(3 + 5 * ch);
        // P3's weight operands are requested now, a whole P2 phase before they are used
        float a3[8][MTO_MAX];
        if (wave < NT3) {
            const float* wo = wpo + (int64_t)(ch * 8) * 64;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int m = 0; m < MTO_MAX; ++m) a3[s][m] = wo[(m < k.MTo ? m : 0) * mts + s * 64] * ((m < k.MTo) ? 1.f : 0.f);
        }
        __syncthreads();
        GD_STAMP(4 + 5 * ch);
        // ---------------- P2: depthwise 3x3 + GELU gate on the slab ----------------
        // wave w owns gate channels {w, w+4, w+8, w+12} of the chunk (wave-uniform -> weights sit in SGPRs); each lane
        // produces 2 horizontally adjacent pixels per (channel, row pair block).
        for (int ci = 0; ci < ((k.dbg & 2) ? 0 : 4); ++ci) {
            const int c16 = __builtin_amdgcn_readfirstlane(wave + 4 * ci);
            const int ca = ch * 16 + c16, cb = k.Hd + ca;
            float wA[9], wB[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) { wA[i] = dww[ca * 9 + i]; wB[i] = dww[cb * 9 + i]; }
            const float ba = dwb ? dwb[ca] : 0.f, bb = dwb ? dwb[cb] : 0.f;
            for (int it = lane; it < TH * (TW / 2); it += 64) {
                const int y = it / (TW / 2), x2 = it - y * (TW / 2);
                float oa[2] = {ba, ba}, ob[2] = {bb, bb};
                const float* pa = hh + c16 * HPp + y * HW2 + 2 * x2;
                const float* pb = pa + 16 * HPp;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float2 a0 = *reinterpret_cast<const float2*>(pa + dy * HW2);
                    const float2 a1 = *reinterpret_cast<const float2*>(pa + dy * HW2 + 2);
                    const float2 b0 = *reinterpret_cast<const float2*>(pb + dy * HW2);
                    const float2 b1 = *reinterpret_cast<const float2*>(pb + dy * HW2 + 2);
                    oa[0] = fmaf(wA[dy * 3], a0.x, fmaf(wA[dy * 3 + 1], a0.y, fmaf(wA[dy * 3 + 2], a1.x, oa[0])));
                    oa[1] = fmaf(wA[dy * 3], a0.y, fmaf(wA[dy * 3 + 1], a1.x, fmaf(wA[dy * 3 + 2], a1.y, oa[1])));
                    ob[0] = fmaf(wB[dy * 3], b0.x, fmaf(wB[dy * 3 + 1], b0.y, fmaf(wB[dy * 3 + 2], b1.x, ob[0])));
                    ob[1] = fmaf(wB[dy * 3], b0.y, fmaf(wB[dy * 3 + 1], b1.x, fmaf(wB[dy * 3 + 2], b1.y, ob[1])));
                }
                *reinterpret_cast<float2*>(gg + c16 * TP + y * TW + 2 * x2) =
                    make_float2(bem_gelu_fast(oa[0]) * ob[0], bem_gelu_fast(oa[1]) * ob[1]);
            }
        }
        GD_STAMP(5 + 5 * ch);
        __syncthreads();
        GD_STAMP(6 + 5 * ch);
        // ---------------- P3: acc3 += W_o[:, 16ch .. 16ch+15] * gg ----------------
        if (wave < NT3 && !(k.dbg & 4)) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float bv = gg[(2 * s + half) * TP + 32 * wave + j];
#pragma unroll
                for (int m = 0; m < MTO_MAX; ++m) acc3[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[s][m], bv, acc3[m], 0, 0, 0);
            }
        }
        GD_STAMP(7 + 5 * ch);
        // hh is rewritten by the next P1 only after every wave passed the barrier that followed P2; gg is rewritten by
        // the next P2 only after the barrier that follows the next P1, which every wave reaches after its P3.
    }

    // ---------------- epilogue: out = acc3 + b_o + x ----------------
    if (wave < NT3) {
        const int pi = 32 * wave + j;
        const int oy = ty0 + pi / TW, ox = tx0 + pi % TW;
        if (oy < k.H && ox < k.W) {
            const float* bpo = k.bpo ? k.bpo + (int64_t)b * k.bpo_bs : nullptr;
            const int64_t pix = (int64_t)oy * k.W + ox;
            float rv[MTO_MAX][16];
#pragma unroll
            for (int m = 0; m < MTO_MAX; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    rv[m][r] = (row < k.C) ? xb[(int64_t)row * HWp + pix] : 0.f;
                }
#pragma unroll
            for (int m = 0; m < MTO_MAX; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row < k.C)
                        k.out[((int64_t)b * k.C + row) * HWp + pix] = acc3[m][r] + (bpo ? bpo[row] : 0.f) + rv[m][r];
                }
        }
    }
    GD_STAMP(90);
}

#undef GD_STAMP

__global__ void pack_gate_kernel(const float* __restrict__ W, float* __restrict__ Wp, int Hd, int K, int KS) {
    // packed tile t (of Hd/16): rows 0-15 = W[16t + r], rows 16-31 = W[Hd + 16t + r - 16]; grid (ceil(per/256), nsets)
    const int MT = Hd / 16;
    const int64_t per = (int64_t)MT * KS * 64;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    const int lane = (int)(i & 63);
    const int64_t t = i >> 6;
    const int st = (int)(t % KS), mt = (int)(t / KS);
    const int r = lane & 31, col = 2 * st + (lane >> 5);
    const int row = (r < 16) ? (16 * mt + r) : (Hd + 16 * mt + r - 16);
    const int set = blockIdx.y;
    Wp[(int64_t)set * per + i] = (col < K) ? W[((int64_t)set * 2 * Hd + row) * K + col] : 0.f;
}

}  // namespace

extern "C" int bem_pack_pw_weight_gate_f32(const float* W, float* Wp, int nsets, int Hd, int K, void* stream) {
    BEM_REQUIRE(W && Wp, "pack_pw_weight_gate: null tensor");
    BEM_REQUIRE(nsets >= 0 && nsets <= 65535 && Hd > 0 && Hd % 16 == 0 && K > 0, "pack_pw_weight_gate: Hd %% 16 != 0 or bad shape");
    if (nsets == 0) return BEM_OK;
    const int KS = cdiv(K, 2);
    dim3 grid((unsigned)cdiv64((int64_t)(Hd / 16) * KS * 64, 256), nsets);
    pack_gate_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(W, Wp, Hd, K, KS);
    return bem_check_launch("pack_pw_weight_gate");
}

extern "C" int bem_gdmlp_fused_f32(const bem_gdmlp_args* a, void* stream) {
    BEM_REQUIRE(a, "gdmlp_fused: null args");
    BEM_REQUIRE(a->x && a->out && a->ln_w && a->ln_b && a->Wpi && a->bpi && a->dww && a->Wpo, "gdmlp_fused: null tensor");
    BEM_REQUIRE(a->B >= 0 && a->B <= 65535 && a->C > 0 && a->C <= 32 * MTO_LIMIT && a->H > 0 && a->W > 0, "gdmlp_fused: bad shape (C <= 160)");
    BEM_REQUIRE(a->Hd > 0 && a->Hd % 16 == 0, "gdmlp_fused: hidden width %d must be a multiple of 16", a->Hd);
    BEM_REQUIRE(a->x != a->out, "gdmlp_fused: in-place not supported (halo reads)");
    if (a->B == 0) return BEM_OK;
    GdK k;
    k.x = a->x; k.out = a->out; k.ln_w = a->ln_w; k.ln_b = a->ln_b; k.eps = a->ln_eps;
    k.Wpi = a->Wpi; k.wpi_bs = a->wpi_bstride; k.bpi = a->bpi; k.bpi_bs = a->bpi_bstride;
    k.dww = a->dww; k.dww_bs = a->dww_bstride; k.dwb = a->dwb; k.dwb_bs = a->dwb_bstride;
    k.Wpo = a->Wpo; k.wpo_bs = a->wpo_bstride; k.bpo = a->bpo; k.bpo_bs = a->bpo_bstride;
    k.C = a->C; k.Hd = a->Hd; k.H = a->H; k.W = a->W; k.KS = cdiv(a->C, 2); k.KSo = a->Hd / 2; k.MTo = cdiv(a->C, 32);
    k.tilesX = cdiv(a->W, TW);
    k.dbg = getenv("BEM_GD_DBG") ? atoi(getenv("BEM_GD_DBG")) : 0;
    static unsigned long long* stamp_buf = nullptr;
    if ((k.dbg & 16) && !stamp_buf) (void)hipMalloc(&stamp_buf, 96 * sizeof(unsigned long long));
    k.stamps = (k.dbg & 16) ? stamp_buf : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int Kp = 2 * k.KS;
    auto lds_bytes = [&](int TH) {
        const int HP = (TH + 2) * HW2, HPp = ((HP + 31) / 32) * 32, TP = TH * TW;
        return ((size_t)Kp * HPp + 32 * HPp + 16 * TP + 2 * HPp + 2 * Kp + 2 * (size_t)a->Hd) * sizeof(float);
    };
    const int TH = (lds_bytes(8) <= 100 * 1024 && a->H > 4) ? 8 : 4;
    const size_t lds = lds_bytes(TH);
    BEM_REQUIRE(lds <= 160 * 1024, "gdmlp_fused: tile does not fit LDS (C = %d)", a->C);
    dim3 grid(k.tilesX * cdiv(a->H, TH), a->B);
#define BEM_GD_LAUNCH(TH_, MTO_)                                                                                              \
    do {                                                                                                                      \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) {                                                                                                      \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gdmlp_fused_kernel<TH_, MTO_>),                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
            attr_set = true;                                                                                                  \
        }                                                                                                                     \
        gdmlp_fused_kernel<TH_, MTO_><<<grid, 256, lds, s>>>(k);                                                               \
    } while (0)
    if (TH == 8) {
        if (k.MTo <= 2) BEM_GD_LAUNCH(8, 2);
        else if (k.MTo <= 3) BEM_GD_LAUNCH(8, 3);
        else BEM_GD_LAUNCH(8, 5);
    } else {
        if (k.MTo <= 2) BEM_GD_LAUNCH(4, 2);
        else if (k.MTo <= 3) BEM_GD_LAUNCH(4, 3);
        else BEM_GD_LAUNCH(4, 5);
    }
#undef BEM_GD_LAUNCH
    if (k.stamps) {
        unsigned long long h[96];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h, stamp_buf, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[gd stamps C=%d] load %llu ln %llu |", a->C, h[1] - h[0], h[2] - h[1]);
        for (int c = 0; c < 3 && c < a->Hd / 16; ++c)
            fprintf(stderr, " ch%d: P1 %llu sync %llu P2 %llu sync %llu P3 %llu |", c, h[3 + 5 * c] - (c ? h[7 + 5 * (c - 1)] : h[2]), h[4 + 5 * c] - h[3 + 5 * c], h[5 + 5 * c] - h[4 + 5 * c], h[6 + 5 * c] - h[5 + 5 * c], h[7 + 5 * c] - h[6 + 5 * c]);
        fprintf(stderr, " total %llu (epilogue %llu)\n", h[90] - h[0], h[90] - h[7 + 5 * (a->Hd / 16 - 1)]);
    }
    return bem_check_launch("gdmlp_fused");
}
