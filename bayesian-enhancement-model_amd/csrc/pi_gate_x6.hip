// gdMlp front half in one kernel:  g = GELU(h1) * h2,  [h1; h2] = dwconv3x3(project_in(LayerNorm2d(x))) + biases
// (reference: basicsr/vmamba/models/vmamba.py:116-131 gdMlp.forward up to the gate, with the block's norm2 :1330).
//
// Why: as three kernels the 8C-channel tensor t = project_in(LN(x)) is written once and read once (level 0 of the bench:
// 1.34 GB each way, 2/3 of the block's HBM traffic).  Here t only ever exists as 32-channel slices of one pixel tile in LDS.
//
// Mapping:
//   * workgroup = 4 waves = one 8 x 32 pixel tile of one image; its 10 x 34 halo is 340 pixels = 11 MFMA pixel blocks of 32.
//     Wave w owns halo blocks w, w + 4, w + 8 and keeps their LayerNorm-ed input, already split into bf16 limbs, in
//     registers for the whole kernel (x is read once per tile; the halo overlap comes from L2).
//   * the 2*Hd project_in rows are packed by the host in "gate order": M-tile j = [16 h1 rows 16j..16j+15 | the 16 h2 rows
//     Hd + 16j ..], so that one 32-row MFMA tile yields both gate inputs of 16 output channels.
//   * per chunk j:  phase A  t(32 rows x 340 halo pixels) = W_j * x on the bf16 matrix cores (exact 3-limb products, see
//     pw_gemm_x6.hip) + bias, forced to zero outside the image (the depthwise conv zero-pads t, not x), into LDS;
//     phase B  wave w takes output rows 2w, 2w+1 (lanes 0-31 / 32-63), lane = column: 3x3 window of (h1, h2) float2 pairs from
//     LDS, both gate inputs advance in one packed FMA against the host-interleaved (w1, w2) depthwise pairs (wave-uniform:
//     scalar loads), erf-form GELU gate, one 128-byte store per half-wave and channel.
//     Measured (64 x 40 x 128 x 128, Hd 160, inside the eval bench): 560 us against 358 + 363 us for project_in + depthwise gate as two
//     kernels, with 2.7 GB less HBM traffic.  The phases do not overlap in practice: with either phase disabled the time drops
//     by that phase's full share (scripts/pig_time.py).  Steps so far: vector loads of the uniform depthwise weights -> scalar
//     loads (820 -> 660 us); all 16 channels of a chunk unrolled (650); (h1, h2) as float2 in LDS + packed FMAs (560).  A form with
//     two pixels per lane and an 8-iteration channel loop had fewer instructions still and was slower (730 us): the kernel is
//     bound by latency at two waves per SIMD, the next lever is occupancy (the resident x limbs are 108 VGPRs).
#include "bem_common.h"
#include "x6_common.h"
#include <stdlib.h>

namespace {

struct PgX {
    const float* x; const float* ln_w; const float* ln_b; float ln_eps;
    const u32x4* Wp;            // gate-order project_in weights, x6-packed: [NCH][KB][3][64]
    int C, Hd, H, W, KB, NCH, tx, dbg;
};

// tile geometry: TH x 32 pixels, NW waves.  <8, 4> for C <= 48: 11 halo blocks, three per wave (their limbs fit the register file), a
// wave owns two rows and all 16 gate channels of a chunk in phase B.  <4, 8> for C <= 80: halo 6 x 34 = 7 blocks, ONE per wave, the
// chunk's packed weights go through LDS, in phase B a wave owns two rows and four gate channels.
constexpr int PG_TW = 32, PG_HW = PG_TW + 2;

// dww / dwb / bpi / g are separate __restrict__ arguments (not struct fields) so that the compiler knows the stores to g cannot
// touch them: the wave-uniform depthwise parameters then come in through scalar loads instead of per-lane vector loads.
template <int KBM, int PG_TH, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void pi_gate_x6_kernel(PgX k, const float* __restrict__ dww, const float* __restrict__ dwb,
                                                            const float* __restrict__ bpi, float bmul, float dbmul, float* __restrict__ g) {
    constexpr int NT = 64 * NW;
    constexpr int PG_NPH = (PG_TH + 2) * PG_HW, PG_NPB = (PG_NPH + 31) / 32, PG_BPW = (PG_NPB + NW - 1) / NW, PG_TS = PG_NPB * 32;
    constexpr int RG = PG_TH / 2;                         // row pairs of the tile = waves per channel group in phase B
    constexpr int NCW = 16 / (NW / RG);                   // gate channels of a chunk per wave in phase B
    static_assert(NW % RG == 0 && 16 % (NW / RG) == 0, "phase-B mapping");
    __shared__ f32x2 T[16 * PG_TS];                       // [gate channel c][halo pixel] = (h1 input, h2 input)
    // NW == 8 (wide inputs): a chunk's packed weights (KBM x 3 KB) go through LDS once per workgroup (fetched during phase B of the
    // previous chunk) instead of 3 KBM registers per lane in every wave
    constexpr bool WLDS = NW == 8;
    constexpr int WSH_N = WLDS ? KBM * 3 * 64 : 1, WPT = (WSH_N + NT - 1) / NT;
    __shared__ u32x4 Wsh[WSH_N];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tyi = tile / k.tx, txi = tile - tyi * k.tx;
    const int y0 = tyi * PG_TH, x0 = txi * PG_TW;
    const int L = k.H * k.W;
    const float* xb = k.x + (int64_t)b * k.C * L;

    // ---- this wave's halo pixel blocks: load, LayerNorm over channels, split into limbs
    float lnw[KBM][8], lnb[KBM][8];
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = 16 * kb + 8 * kh + e;
            const float on = ch < k.C ? 1.f : 0.f;
            lnw[kb][e] = k.ln_w[min(ch, k.C - 1)] * on;
            lnb[kb][e] = k.ln_b[min(ch, k.C - 1)] * on;
        }
    u32x4 xl[PG_BPW][KBM][3];
    float msk[PG_BPW];
    int hpo[PG_BPW];
#pragma unroll
    for (int i = 0; i < PG_BPW; ++i) {
        const int hp = min((wave + NW * i) * 32 + n, PG_TS - 1);
        hpo[i] = hp;
        const int hy = hp / PG_HW, hx = hp - hy * PG_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool in = hp < PG_NPH && gy >= 0 && gy < k.H && gx >= 0 && gx < k.W;
        msk[i] = in ? 1.f : 0.f;
        const int off = min(max(gy, 0), k.H - 1) * k.W + min(max(gx, 0), k.W - 1);
        float xr[KBM][8];
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 16 * kb + 8 * kh + e;
                const float v = xb[(int64_t)min(ch, k.C - 1) * L + off];
                xr[kb][e] = ch < k.C ? v : 0.f;
            }
        const float inv = 1.f / (float)k.C;
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += xr[kb][e];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * inv;
        float q = 0.f;
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (16 * kb + 8 * kh + e < k.C) ? xr[kb][e] - mean : 0.f;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = 1.f / sqrtf(q * inv + k.ln_eps);
#pragma unroll
        for (int kb = 0; kb < KBM; ++kb) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (xr[kb][e] - mean) * rstd * lnw[kb][e] + lnb[kb][e];
            split8(v, xl[i][kb][0], xl[i][kb][1], xl[i][kb][2]);
        }
    }

    const u32x4* wbase = k.Wp + lane;
    const int64_t ch_stride = (int64_t)k.KB * 3 * 64;
    u32x4 wc[WLDS ? 1 : KBM][3];
    u32x4 wst[WPT];                                       // WLDS: this thread's share of the next chunk's weights on their way to LDS
    float4 bq[4];
    auto fetch_wsh = [&](int j) {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int i = min((int)threadIdx.x + NT * u, WSH_N - 1);
            const int kb = i / 192, rest = i - kb * 192;                   // [kb][limb][lane]
            const uint32_t mk = kb < k.KB ? 0xffffffffu : 0u;
            const u32x4 w = k.Wp[(int64_t)j * ch_stride + (int64_t)min(kb, k.KB - 1) * 192 + rest];
            wst[u] = u32x4{w[0] & mk, w[1] & mk, w[2] & mk, w[3] & mk};
        }
    };
    auto store_wsh = [&]() {
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int i = (int)threadIdx.x + NT * u;
            if (i < WSH_N) Wsh[i] = wst[u];
        }
    };
    auto load_w = [&](int j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(bpi + 32 * j + 8 * q + 4 * kh);
            bq[q] = make_float4(v.x * bmul, v.y * bmul, v.z * bmul, v.w * bmul);
        }
        if (WLDS) { fetch_wsh(j); return; }
#pragma unroll
        for (int kb = 0; kb < (WLDS ? 0 : KBM); ++kb) {
            const uint32_t mk = kb < k.KB ? 0xffffffffu : 0u;
            const u32x4* wp = wbase + (int64_t)j * ch_stride + (int64_t)min(kb, k.KB - 1) * 3 * 64;
#pragma unroll
            for (int li = 0; li < 3; ++li) {
                const u32x4 w = wp[li * 64];
                wc[kb][li] = u32x4{w[0] & mk, w[1] & mk, w[2] & mk, w[3] & mk};
            }
        }
    };
    load_w(0);
    if (WLDS) { store_wsh(); __syncthreads(); }

    // phase-B geometry: output pixel (ry, n) of the tile; its window starts at halo (ry, n).  Wave w owns the row pair w % RG and the
    // gate channels NCW (w / RG) .. of every chunk
    const int ry = (NW == RG) ? 2 * wave + kh : 2 * (wave % RG) + kh;
    const int c_lo = (NW == RG) ? 0 : NCW * __builtin_amdgcn_readfirstlane(wave / RG);
    const int oy = y0 + ry, ox = x0 + n;
    const bool ost = oy < k.H && ox < k.W;
    const int wbase_lds = ry * PG_HW + n;
    float* gb = g + (int64_t)b * k.Hd * L + (int64_t)min(oy, k.H - 1) * k.W + min(ox, k.W - 1);

    for (int j = 0; j < k.NCH; ++j) {
        // ---- phase A
#pragma unroll
        for (int i = 0; i < PG_BPW; ++i) {
            if (wave + NW * i < PG_NPB && !(k.dbg & 1)) {                           // wave-uniform
                f32x16 hi, lo;
#pragma unroll
                for (int r = 0; r < 16; ++r) hi[r] = lo[r] = 0.f;
                if (WLDS) {
#pragma unroll
                    for (int kb = 0; kb < KBM; ++kb) {
                        const u32x4* wp = Wsh + kb * 192 + lane;
                        const u32x4 wl[3] = {wp[0], wp[64], wp[128]};
                        mac6(wl, xl[i][kb], hi, lo);
                    }
                } else {
#pragma unroll
                    for (int kb = 0; kb < (WLDS ? 0 : KBM); ++kb) mac6(wc[kb], xl[i][kb], hi, lo);
                }
                f32x2* tp = T + 4 * kh * PG_TS + hpo[i];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // accumulator rows 8q + e + 4kh (h1 input of gate channel c = 8q + e + 4kh) and 16 + the same (its h2 input)
                        const int r = 4 * q + e;
                        const float4 b1 = bq[q], b2 = bq[q + 2];
                        const float v1 = e == 0 ? b1.x : e == 1 ? b1.y : e == 2 ? b1.z : b1.w;
                        const float v2 = e == 0 ? b2.x : e == 1 ? b2.y : e == 2 ? b2.z : b2.w;
                        tp[(8 * q + e) * PG_TS] = f32x2{(hi[r] + lo[r] + v1) * msk[i], (hi[r + 8] + lo[r + 8] + v2) * msk[i]};
                    }
            }
        }
        __syncthreads();
        load_w(min(j + 1, k.NCH - 1));                            // in flight during phase B
        // ---- phase B: all 16 channel pairs unrolled -- the scalar weight loads, LDS window reads and the exp / rcp chains of
        // different channels are independent, and with two waves per SIMD that ILP is what hides their latencies
        if (!(k.dbg & 2)) {
            float hg[NCW], hv[NCW];
#pragma unroll
            for (int cw = 0; cw < NCW; ++cw) {
                const int c = c_lo + cw;
                const int cg = 16 * j + c;
                const f32x2* wq = reinterpret_cast<const f32x2*>(dww) + cg * 9;      // host-interleaved (w1, w2) per tap
                f32x2 acc = reinterpret_cast<const f32x2*>(dwb)[cg] * dbmul;
                const f32x2* tp = T + c * PG_TS + wbase_lds;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc = __builtin_elementwise_fma(wq[3 * dy + dx], tp[dy * PG_HW + dx], acc);
                hg[cw] = acc[0]; hv[cw] = acc[1];
            }
#pragma unroll
            for (int cw = 0; cw < NCW; ++cw) hg[cw] = bem_gelu_fast(hg[cw]) * hv[cw];
            if (ost && !(k.dbg & 4)) {
#pragma unroll
                for (int cw = 0; cw < NCW; ++cw) gb[(int64_t)(16 * j + c_lo + cw) * L] = hg[cw];
            }
        }
        if (WLDS) store_wsh();                                    // phase A of this chunk is behind barrier 1: Wsh is free
        __syncthreads();
    }
}

}  // namespace

extern "C" int bem_pi_gate_x6_f32(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const float* Wp_gate,
                                  const float* bias_gate, const float* dw_gate, const float* dwb_gate, float* g, int B, int C, int Hd,
                                  int H, int W, void* stream) {
    BEM_REQUIRE(x && ln_w && ln_b && Wp_gate && dw_gate && g, "pi_gate_x6: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && C > 0 && C <= 80 && Hd > 0 && Hd % 16 == 0 && H > 0 && W > 0,
                "pi_gate_x6: needs C <= 80 and Hd %% 16 == 0 (got C = %d, Hd = %d)", C, Hd);
    BEM_REQUIRE(((uintptr_t)Wp_gate & 15) == 0 && (!bias_gate || ((uintptr_t)bias_gate & 15) == 0) && ((uintptr_t)dw_gate & 15) == 0,
                "pi_gate_x6: packed weights, bias and depthwise weights must be 16-byte aligned");
    BEM_REQUIRE((int64_t)2 * Hd * H * W < (1ll << 31), "pi_gate_x6: plane set too large for 32-bit offsets");
    if (B == 0) return BEM_OK;
    PgX k;
    k.x = x; k.ln_w = ln_w; k.ln_b = ln_b; k.ln_eps = ln_eps; k.Wp = reinterpret_cast<const u32x4*>(Wp_gate);
    // absent biases: read the (always present, >= 2 Hd floats) depthwise weights instead and multiply by zero -- no branch next to a load
    const float* bpi = bias_gate ? bias_gate : dw_gate;
    const float* dwbp = dwb_gate ? dwb_gate : dw_gate;
    const float bmul = bias_gate ? 1.f : 0.f, dbmul = dwb_gate ? 1.f : 0.f;
    k.C = C; k.Hd = Hd; k.H = H; k.W = W; k.KB = cdiv(C, 16); k.NCH = Hd / 16; k.tx = cdiv(W, PG_TW);
    k.dbg = getenv("BEM_PIG_DBG") ? atoi(getenv("BEM_PIG_DBG")) : 0;
    const int th = k.KB <= 3 ? 8 : 4;
    dim3 grid(k.tx * cdiv(H, th), 1, B);
    hipStream_t s = (hipStream_t)stream;
    if (k.KB <= 1) pi_gate_x6_kernel<1, 8, 4><<<grid, 256, 0, s>>>(k, dw_gate, dwbp, bpi, bmul, dbmul, g);
    else if (k.KB == 2) pi_gate_x6_kernel<2, 8, 4><<<grid, 256, 0, s>>>(k, dw_gate, dwbp, bpi, bmul, dbmul, g);
    else if (k.KB == 3) pi_gate_x6_kernel<3, 8, 4><<<grid, 256, 0, s>>>(k, dw_gate, dwbp, bpi, bmul, dbmul, g);
    else if (k.KB == 4) pi_gate_x6_kernel<4, 4, 8><<<grid, 512, 0, s>>>(k, dw_gate, dwbp, bpi, bmul, dbmul, g);
    else pi_gate_x6_kernel<5, 4, 8><<<grid, 512, 0, s>>>(k, dw_gate, dwbp, bpi, bmul, dbmul, g);
    return bem_check_launch("pi_gate_x6");
}
