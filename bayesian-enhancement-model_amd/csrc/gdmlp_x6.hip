// The whole gdMlp branch of a VSSBlock in one kernel:
//     out = x + W_o * (GELU(h1) * h2) + b_o,     [h1; h2] = dwconv3x3(W_i * LayerNorm2d(x) + b_i) + b_dw
// (reference: basicsr/vmamba/models/vmamba.py:116-133 gdMlp.forward, with the block's norm2 and residual :1330-1333).
//
// Why: as separate kernels the 8C-channel tensor t = project_in(LN(x)) and the 4C-channel gate tensor g are each written once and
// read once -- 2 x (8C + 4C) . P . 4 bytes against the 2 . C . P . 4 bytes of x in / out that the branch needs (12x).  Here x is read
// once (plus the halo rows from L2), out is written once, t exists as 32-row slices of one pixel tile in LDS and g as a 16-channel
// slice that goes straight back into the matrix cores as the K-slice of project_out.
//
// Mapping (one workgroup = 4 waves = one 4 x 32 pixel tile of one image; 2 workgroups per CU):
//   * the tile's 6 x 34 halo is 204 pixels = 7 MFMA pixel blocks of 32; wave w owns blocks w and w + 4 and keeps their LayerNorm-ed
//     input, split into three bf16 limbs (x6_common.h), in registers for the whole kernel.
//   * project_in rows are packed by the host in gate-interleaved order: M-tile j row 2c + s = W_i[s Hd + 16 j + c], so that accumulator
//     register pair (2i, 2i + 1) of a lane is the (h1, h2) input of ONE gate channel -- the float2 that phase B consumes.
//   * per chunk j of 16 gate channels:
//       phase A  t (32 rows x 204 halo pixels) = W_j xn on the bf16 matrix cores (six exact limb products) + bias, zero outside the
//                image (the depthwise conv zero-pads t, not x), into LDS as [channel][halo pixel] float2;
//       phase B  wave w takes gate channels 4w .. 4w+3 for ALL 128 pixels (lane = column, two rows per lane): the 18 depthwise weights
//                of a channel pair are wave-uniform scalars used by both pixels, the 4 x 3 window is read once for the two pixels;
//                packed FMAs on (h1, h2), erf-form GELU evaluated two pixels at a time; g -> LDS as [pixel][16 channels];
//       phase C  (runs with phase A of the NEXT chunk: both are matrix-core work) wave w takes pixel row w: its lanes read back the
//                8 k-values of the B operand, split them into limbs and accumulate out[C x 32 px] += W_o[:, chunk] g.
//     The chunk's parameters (packed W_i, W_o, bias pairs, depthwise taps + bias) travel global -> LDS by LDS-DMA: no VGPRs, one copy
//     per workgroup instead of one per wave, double-buffered and requested one whole iteration ahead of their use.  Nothing in the
//     chunk loop is fetched by a scalar or vector load.  The LDS regions are separate __shared__ objects, so the compiler may order
//     the bias reads of phase A ahead of its T stores.
//   * epilogue: + b_o + x (residual), 128-byte row segments per half-wave.
// Two barriers per chunk.
#include "bem_common.h"
#include "x6_common.h"

namespace {

struct GdX {
    const float* x; const float* ln_w; const float* ln_b; float ln_eps;
    const u32x4* Wpi;           // gate-interleaved project_in weights, x6-packed: [NCH][KB][3][64]
    const u32x4* Wpo;           // project_out weights, x6-packed [MT][NCH][3][64]
    const float* bgi;           // project_in bias as (h1, h2) pairs per gate channel: [NCH][16][2]  (zeros when the layer has no bias)
    const float* dw10;          // depthwise taps + bias per gate channel: [Hd][10] (w1, w2) pairs, slot 9 = (b1, b2)
    int C, Hd, H, W, NCH, tx;
};

constexpr int GD_TH = 4, GD_TW = 32, GD_HW = GD_TW + 2;
constexpr int GD_NPH = (GD_TH + 2) * GD_HW;          // 204 halo pixels
constexpr int GD_NPB = (GD_NPH + 31) / 32;           // 7 blocks
constexpr int GD_TS = 208;                           // row stride of T (>= 205: the clamp slot of the unused lanes of block 6)
constexpr int GD_GS = 20;                            // dwords per pixel row of G (16 + 4: conflict-free b128 writes and reads)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// GELU(a) * b (erf by Abramowitz & Stegun 7.1.26 on the hardware rcp / exp2, as bem_gelu_fast).  Scalar f32 instructions on purpose:
// on gfx950 a v_pk_*_f32 instruction does not overlap with another wave's MFMAs on the same SIMD, plain v_fma_f32 / v_exp_f32 do
// (scripts/probes/coexec_probe.hip: 26 % of a packed stream hidden behind a matrix stream, 77 % of a scalar one, exp2 / rcp entirely) --
// and both forms run at 64 FLOP per clock and SIMD, so nothing is lost by not packing.  This file is compiled with -fno-slp-vectorize.
__device__ __forceinline__ float gelu_gate1(float a, float b) {
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(a), 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(a * a * -0.72134752044448170368f);           // exp(-(a / sqrt 2)^2)
    const float r = fmaf(-(p * t), e, 1.f);                                              // erf(|a| / sqrt 2)
    const float h = 0.5f * a;
    return fmaf(h, copysignf(r, a), h) * b;
}

template <int KBM, int MTO, int WOB, bool PL2>
__global__ __launch_bounds__(256, 2) void gdmlp_x6_kernel(GdX k, const float* __restrict__ bpo, float bomul, float* __restrict__ out) {
    constexpr int NPI = 3 * KBM, NPO = 3 * MTO;                                        // 1 KiB pieces of a W_i / W_o chunk
    __shared__ __attribute__((aligned(16))) f32x2 T[16 * GD_TS];                       // [gate channel c][halo pixel] = (h1 input, h2 input)
    __shared__ __attribute__((aligned(16))) float G[128 * GD_GS];                      // [tile pixel][16 gate channels (+4 pad)]
    __shared__ __attribute__((aligned(16))) u32x4 Wis[2][NPI * 64];                    // W_i chunk [kb][limb][lane], chunk j in buffer j & 1
    __shared__ __attribute__((aligned(16))) u32x4 Wos[WOB][NPO * 64];                  // W_o chunk [mt][limb][lane], chunk j in buffer j & (WOB - 1)
    __shared__ __attribute__((aligned(16))) f32x2 Bss[2][16];                          // project_in bias pairs of the chunk
    __shared__ __attribute__((aligned(16))) f32x2 DWs[2][16 * 10];                     // depthwise taps + bias of the chunk
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int tyi = tile / k.tx, txi = tile - tyi * k.tx;
    const int y0 = tyi * GD_TH, x0 = txi * GD_TW;
    const int L = k.H * k.W;
    const float* xb = k.x + (int64_t)b * k.C * L;

    // Chunk parameters by LDS-DMA, requested a whole iteration before their first use (the wait in front of an iteration's last barrier
    // then finds them landed: waiting for a request made in the same half-iteration exposes ~1 us of loaded L2 latency per chunk, which
    // was 95 % of this kernel's time).  Piece p of a chunk is moved by wave p % 4; all selects are scalar.
    const uint32_t voff = 16 * lane;
    const uint32_t wi_lds = lds_addr(Wis), wo_lds = lds_addr(Wos), bs_lds = lds_addr(Bss), dw_lds = lds_addr(DWs);
    auto dma_in = [&](int ji, int buf) {                                               // W_i, bias pairs, depthwise parameters of chunk ji
#pragma unroll
        for (int t = 0; t < (NPI + 3) / 4; ++t) {
            const int p = wave + 4 * t;
            if (p < NPI) glds16(k.Wpi + ((int64_t)ji * NPI + p) * 64, voff, wi_lds + (buf * NPI + p) * 1024);
        }
        if (wave == (NPI & 3)) {                                                       // the wave with the fewest W_i pieces
            const float* src = k.dw10 + (int64_t)ji * 320;                             // 1280 bytes = one full piece + 16 lanes
            glds16(src, voff, dw_lds + buf * 1280);
            if (lane < 16) glds16(src + 256, voff, dw_lds + buf * 1280 + 1024);
        }
        if (wave == ((NPI + 1) & 3) && lane < 8) glds16(k.bgi + (int64_t)ji * 32, voff, bs_lds + buf * 128);
    };
    auto dma_out = [&](int jo, int buf) {                                              // W_o chunk jo: MTO row blocks of three limbs
#pragma unroll
        for (int t = 0; t < (NPO + 3) / 4; ++t) {
            const int p = wave + 4 * t, mt = p / 3, li = p - 3 * mt;
            if (p < NPO) glds16(k.Wpo + (((int64_t)mt * k.NCH + jo) * 3 + li) * 64, voff, wo_lds + (buf * NPO + p) * 1024);
        }
    };
    dma_in(0, 0);

    // phase-C geometry: wave = tile row, lane (n, kh) = pixel n, k-half kh
    const int oy = y0 + wave, ox = x0 + n;
    // residual + output bias of this wave's pixel row (rows (r & 3) + 8 (r >> 2) + 4 kh of each M-tile): requested
    // ahead of the last phase C
    f32x16 rs[MTO];
    const bool opix = oy < k.H && ox < k.W;
    const int64_t po = (int64_t)min(oy, k.H - 1) * k.W + min(ox, k.W - 1);
    auto load_res = [&]() {
        const float* rb = xb + po;
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = min(32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh, k.C - 1);
                rs[mt][r] = fmaf(bpo[row], bomul, rb[(int64_t)row * L]);
            }
    };

    // ---- this wave's halo pixel blocks (w, w + 4): load, LayerNorm over channels, zero outside the image, split into limbs
    u32x4 xl[2][KBM][3];
    float msk[2];
    uint32_t mbit[2];                  // all ones inside the image
    int hpo[2];
    {
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
        // PL2: every global load of the prologue is requested before the first value is used -- one exposed memory latency per workgroup,
        // not one per halo block (C = 80 has no registers for that: 2 x 40 raw values next to 120 limb registers)
        float xr[2][KBM][8];
        auto load_block = [&](int i) {
            const int hp = min((wave + 4 * i) * 32 + n, GD_TS - 1);
            hpo[i] = hp;
            const int hy = hp / GD_HW, hx = hp - hy * GD_HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = hp < GD_NPH && gy >= 0 && gy < k.H && gx >= 0 && gx < k.W && (wave + 4 * i) < GD_NPB;
            msk[i] = in ? 1.f : 0.f;
            mbit[i] = in ? 0xffffffffu : 0u;
            const int off = min(max(gy, 0), k.H - 1) * k.W + min(max(gx, 0), k.W - 1);
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 16 * kb + 8 * kh + e;
                    const float v = xb[(int64_t)min(ch, k.C - 1) * L + off];
                    xr[i][kb][e] = ch < k.C ? v : 0.f;
                }
        };
        auto norm_block = [&](int i) {
            const float inv = 1.f / (float)k.C;
            float s = 0.f;
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += xr[i][kb][e];
            s += __shfl_xor(s, 32, 64);
            const float mean = s * inv;
            float q = 0.f;
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (16 * kb + 8 * kh + e < k.C) ? xr[i][kb][e] - mean : 0.f;
                    q = fmaf(d, d, q);
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = msk[i] / sqrtf(q * inv + k.ln_eps);
#pragma unroll
            for (int kb = 0; kb < KBM; ++kb) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((xr[i][kb][e] - mean) * rstd) * lnw[kb][e] + lnb[kb][e] * msk[i];
                split8(v, xl[i][kb][0], xl[i][kb][1], xl[i][kb][2]);
            }
        };
        if (PL2) { load_block(0); load_block(1); norm_block(0); norm_block(1); }
        else { load_block(0); norm_block(0); load_block(1); norm_block(1); }
    }

    // phase-B geometry: lane = column n, tile rows 2 kh and 2 kh + 1; the window of both starts at halo (2 kh, n)
    const int wb_lds = 2 * kh * GD_HW + n;
    const int c_lo = 4 * wave;

    f32x16 oh[MTO];
#pragma unroll
    for (int mt = 0; mt < MTO; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oh[mt][r] = 0.f;

    auto phase_c = [&](const u32x4* Wo) {
        const float* gp = G + (wave * 32 + n) * GD_GS + 8 * kh;
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gp), gb = *reinterpret_cast<const f32x4*>(gp + 4);
        const float v[8] = {ga[0], ga[1], ga[2], ga[3], gb[0], gb[1], gb[2], gb[3]};
        u32x4 gl[3];
        split8(v, gl[0], gl[1], gl[2]);
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt) {
            const u32x4* wp = Wo + mt * 192 + lane;
            const u32x4 wl[3] = {wp[0], wp[64], wp[128]};
            f32x16 lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) lo[r] = 0.f;
            mac6(wl, gl, oh[mt], lo);
            oh[mt] += lo;
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma clang loop unroll(disable)
    for (int j = 0; j < k.NCH; ++j) {
        const int cur = j & 1;
        // requests for the next iteration (W_i, bias, depthwise parameters of chunk j + 1) and, with two W_o buffers, for phase C of this
        // chunk: their buffers were last read before the previous iteration's final barrier
        if (j + 1 < k.NCH) dma_in(j + 1, cur ^ 1);
        if (WOB == 2) dma_out(j, cur);
        const u32x4* const Wi = Wis[cur];
        // ---- phase A (chunk j) and phase C (chunk j - 1): matrix-core work
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (wave + 4 * i < GD_NPB) {                                               // wave-uniform
                // accumulator rows 2(q&1) + 8(q>>1) + 4kh (+1) are the (h1, h2) inputs of gate channel c = (q&1) + 4(q>>1) + 2kh:
                // the small-product accumulator starts from the channel's bias pair (zero outside the image: the conv pads t with zeros)
                f32x16 hi, lo;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x2 bp = Bss[cur][(q & 1) + 4 * (q >> 1) + 2 * kh];
                    lo[2 * q] = bitsf(fbits(bp[0]) & mbit[i]);
                    lo[2 * q + 1] = bitsf(fbits(bp[1]) & mbit[i]);
                    hi[2 * q] = hi[2 * q + 1] = 0.f;
                }
#pragma unroll
                for (int kb = 0; kb < KBM; ++kb) {
                    const u32x4* wp = Wi + kb * 192 + lane;
                    const u32x4 wl[3] = {wp[0], wp[64], wp[128]};
                    hi = mfma16(wl[0], xl[i][kb][0], hi);
                    lo = mfma16(wl[0], xl[i][kb][2], lo);
                    lo = mfma16(wl[2], xl[i][kb][0], lo);
                    lo = mfma16(wl[1], xl[i][kb][1], lo);
                    lo = mfma16(wl[0], xl[i][kb][1], lo);
                    lo = mfma16(wl[1], xl[i][kb][0], lo);
                }
                f32x2* tp = T + 2 * kh * GD_TS + hpo[i];
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    tp[((q & 1) + 4 * (q >> 1)) * GD_TS] = f32x2{hi[2 * q] + lo[2 * q], hi[2 * q + 1] + lo[2 * q + 1]};
            }
        }
        if (j) phase_c(Wos[(j - 1) & (WOB - 1)]);
        __syncthreads();
        // ---- phase B (chunk j); with one W_o buffer its chunk is requested now that phase C has released the buffer
        if (WOB == 1) dma_out(j, 0);
        {
            f32x2 g[4];                                                                // [channel] = (pixel row 2kh, pixel row 2kh + 1)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const f32x4* wv = reinterpret_cast<const f32x4*>(DWs[cur] + (c_lo + cc) * 10);   // wave-uniform address: broadcast reads
                const f32x4 w01 = wv[0], w23 = wv[1], w45 = wv[2], w67 = wv[3], w8b = wv[4];
                const f32x2 wq[9] = {{w01[0], w01[1]}, {w01[2], w01[3]}, {w23[0], w23[1]}, {w23[2], w23[3]}, {w45[0], w45[1]},
                                     {w45[2], w45[3]}, {w67[0], w67[1]}, {w67[2], w67[3]}, {w8b[0], w8b[1]}};
                const f32x2 bias = {w8b[2], w8b[3]};
                const f32x2* tp = T + (c_lo + cc) * GD_TS + wb_lds;
                f32x2 win[4][3];
#pragma unroll
                for (int dy = 0; dy < 4; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) win[dy][dx] = tp[dy * GD_HW + dx];
                float a0x = bias[0], a0y = bias[1], a1x = bias[0], a1y = bias[1];         // (h1, h2) of pixel rows 2kh and 2kh + 1
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) {
                        const f32x2 w = wq[3 * ty + tx], u0 = win[ty][tx], u1 = win[ty + 1][tx];
                        a0x = fmaf(w[0], u0[0], a0x); a0y = fmaf(w[1], u0[1], a0y);
                        a1x = fmaf(w[0], u1[0], a1x); a1y = fmaf(w[1], u1[1], a1y);
                    }
                g[cc] = f32x2{gelu_gate1(a0x, a0y), gelu_gate1(a1x, a1y)};
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
                *reinterpret_cast<f32x4*>(G + ((2 * kh + p) * 32 + n) * GD_GS + c_lo) = f32x4{g[0][p], g[1][p], g[2][p], g[3][p]};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    load_res();                                                                        // requested ahead of the last matrix phase
    phase_c(Wos[(k.NCH - 1) & (WOB - 1)]);

    // ---- epilogue: + bias + residual, rows (r & 3) + 8 (r >> 2) + 4 kh of each M-tile, 128-byte segments per half-wave
    if (opix) {
        float* ob = out + (int64_t)b * k.C * L + po;
#pragma unroll
        for (int mt = 0; mt < MTO; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < k.C) ob[(int64_t)row * L] = oh[mt][r] + rs[mt][r];
            }
    }
}

}  // namespace

extern "C" int bem_gdmlp_x6_f32(const float* x, const float* ln_w, const float* ln_b, float ln_eps, const float* Wp_gate,
                                const float* bias_gate, const float* dw_gate10, const float* Wp_out,
                                const float* bias_out, float* out, int B, int C, int Hd, int H, int W, void* stream) {
    BEM_REQUIRE(x && ln_w && ln_b && Wp_gate && bias_gate && dw_gate10 && Wp_out && out, "gdmlp_x6: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && C > 0 && C <= 80 && Hd > 0 && Hd % 16 == 0 && H > 0 && W > 0,
                "gdmlp_x6: needs C <= 80 and Hd %% 16 == 0 (got C = %d, Hd = %d)", C, Hd);
    BEM_REQUIRE((((uintptr_t)Wp_gate | (uintptr_t)Wp_out | (uintptr_t)bias_gate | (uintptr_t)dw_gate10) & 15) == 0,
                "gdmlp_x6: packed weights, bias pairs and depthwise parameters must be 16-byte aligned");
    BEM_REQUIRE(x != out, "gdmlp_x6: in-place operation is not supported (halo reads)");
    BEM_REQUIRE((int64_t)C * H * W < (1ll << 31), "gdmlp_x6: plane set too large for 32-bit offsets");
    if (B == 0) return BEM_OK;
    GdX k;
    k.x = x; k.ln_w = ln_w; k.ln_b = ln_b; k.ln_eps = ln_eps;
    k.Wpi = reinterpret_cast<const u32x4*>(Wp_gate); k.Wpo = reinterpret_cast<const u32x4*>(Wp_out); k.bgi = bias_gate;
    k.dw10 = dw_gate10;
    k.C = C; k.Hd = Hd; k.H = H; k.W = W; k.NCH = Hd / 16; k.tx = cdiv(W, GD_TW);
    // absent output bias: read an always-present array and multiply by zero -- no branch next to a load
    const float* bpop = bias_out ? bias_out : ln_w;
    const float bomul = bias_out ? 1.f : 0.f;
    dim3 grid(k.tx * cdiv(H, GD_TH), 1, B);
    hipStream_t s = (hipStream_t)stream;
    const int KB = cdiv(C, 16);
    // two workgroups per CU in every variant (LDS: T 26 KB + G 10 KB + two W_i buffers + one or two W_o buffers <= 80 KB)
    if (KB == 1) gdmlp_x6_kernel<1, 1, 2, true><<<grid, 256, 0, s>>>(k, bpop, bomul, out);
    else if (KB == 2) gdmlp_x6_kernel<2, 1, 2, true><<<grid, 256, 0, s>>>(k, bpop, bomul, out);
    else if (KB == 3) gdmlp_x6_kernel<3, 2, 2, true><<<grid, 256, 0, s>>>(k, bpop, bomul, out);
    else if (KB == 4) gdmlp_x6_kernel<4, 2, 2, true><<<grid, 256, 0, s>>>(k, bpop, bomul, out);
    else gdmlp_x6_kernel<5, 3, 1, false><<<grid, 256, 0, s>>>(k, bpop, bomul, out);
    return bem_check_launch("gdmlp_x6");
}
