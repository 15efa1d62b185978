// Dense convolutions in "row form" on the bf16-limb matrix-core machinery of pw_gemm_x6.hip: the 4x4 stride-2 pad-1 down-sampling convs of
// the U-Nets (DecompDualBranchDDWavelet_arch.py:40-41) and the 3x3 stride-1 pad-1 convs of the decomposition nets and the first / last layers
// (basicsr/QD/model4.py:181-200, DecompDualBranchDDWavelet_arch.py:190), written for what bounds them:
//     out[co][yo][xo] = act(bias[co] + sum_{ci, ky, kx} W[co][ci][ky][kx] x[ci][S yo - 1 + ky][S xo - 1 + kx]) + res1 + res2
//   * a lane owns NPX = 4 / S neighbouring output pixels of a row: for one input row they need six input columns, 4 S m' - 1 .. + 4.
//     Four of them are ONE aligned 16-byte load per channel (a half-wave reads 512 contiguous bytes of a plane row); the two outer ones are
//     the neighbour lanes' values, moved by DPP wave shifts and zeroed at the row ends, which is exactly the zero padding.  The shifted-tap
//     form this replaces (conv_taps_x6_kernel) issues KS x KS x 16 scalar loads per k-block where this one issues KS x 8 vector loads.
//   * the six columns are split into bf16 limbs once (x6_common.h) and serve all kx taps of all NPX pixels: 6 splits for 8 (4x4) / 12 (3x3)
//     tap uses.
//   * tap weights are KS^2 x the bytes of a 1x1 layer; fetched per wave from L2 they would need ~20 TB/s.  A workgroup stages the
//     3 KS MTW 1-KiB operand blocks of a step (k-block, input row) in LDS by LDS-DMA, double-buffered, requested one step ahead.
//   * the input rows of the next step are requested (into registers) before the matrix work of the current one.
// One workgroup = 4 waves = 128 NPX consecutive output pixels x MTW row blocks of 32 output channels; two workgroups per CU.
// Shapes: Wo / NPX (the lanes of an output row) a power of two <= 32, so that no row crosses a half-wave: Wo in {2 .. 64} for the 4x4 stride-2
// form (W = 2 Wo, H = 2 Ho), W in {4 .. 128} for 3x3; Cin % 8 == 0.  Wp = bem_pack_pw_weight_x6 of the (KS^2, Cout, Cin) tap matrices,
// tap = KS ky + kx -- the format of bem_conv4x4s2_x6_f32 / bem_conv3x3_x6_f32.
#include "bem_common.h"
#include "scan_common.h"
#include "x6_common.h"

namespace {

struct CrX {
    const float* x; int64_t x_bs;
    const u32x4* Wp; const float* bias; const float* res1; const float* res2; float* out;
    int Cin, H, W, Ho, Wo, Cout, KB, MT, wo_shift, relu, mt_first;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int MTW, int KS, int S>
__global__ __launch_bounds__(256, 2) void conv_rows_x6_kernel(CrX k) {
    constexpr int NPX = 4 / S;                                                         // output pixels of a lane
    constexpr int NP = 3 * KS * MTW;                                                   // 1 KiB pieces of a step: [kx][m][limb]
    __shared__ __attribute__((aligned(16))) u32x4 Ws[2][NP * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), kh = lane >> 5, n = lane & 31;
    const int b = blockIdx.z, mt0 = k.mt_first + blockIdx.y * MTW;
    const int Lo = k.Ho * k.Wo, Li = k.H * k.W;
    const int p0 = (xcd_tile(blockIdx.x, gridDim.x) * 4 + wave) * (32 * NPX);
    const int p = p0 + NPX * n;
    const bool live = p < Lo;
    const int pc = live ? p : 0;
    const int yo = pc >> k.wo_shift, xo = pc & (k.Wo - 1);
    const float lmask = xo == 0 ? 0.f : 1.f, rmask = xo == k.Wo - NPX ? 0.f : 1.f;      // the row's first / last lane: padding columns
    const float* xb = k.x + (int64_t)b * k.x_bs;
    const uint32_t voff = 16 * lane, ws_lds = lds_addr(Ws);
    const int NS = KS * k.KB;                                                          // step s = KS kb + r (input row S yo - 1 + r)

    auto dma_w = [&](int s, int buf) {
        const int kb = s / KS, r = s - kb * KS;
#pragma unroll
        for (int t = 0; t < (NP + 3) / 4; ++t) {
            const int pi = wave + 4 * t, kx = pi / (3 * MTW), m = (pi / 3) % MTW, li = pi % 3;
            if (pi < NP)
                glds16(k.Wp + ((((int64_t)(KS * r + kx) * k.MT + (mt0 + m)) * k.KB + kb) * 3 + li) * 64, voff, ws_lds + (buf * NP + pi) * 1024);
        }
    };
    // the 8 channels (16 kb + 8 kh + e) of this lane at input row S yo - 1 + r, own columns S xo .. S xo + 3; rows outside the image: clamped
    // address, zero mask.  Past Cin (a half-filled last k-block) a valid channel is re-read: the packed weights there are zero.
    auto load_x = [&](int s, f32x4v (&dst)[8], float& mk) {
        const int kb = min(s / KS, k.KB - 1), r = s - (s / KS) * KS;
        const int yi = S * yo - 1 + r;
        mk = (live && yi >= 0 && yi < k.H) ? 1.f : 0.f;
        const int off = min(max(yi, 0), k.H - 1) * k.W + S * xo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = min(16 * kb + 8 * kh + e, k.Cin - 1);
            dst[e] = *reinterpret_cast<const f32x4v*>(xb + (int64_t)ch * Li + off);
        }
    };

    f32x16 acc[MTW][NPX], alo[MTW][NPX];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NPX; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = alo[m][t][r] = 0.f;

    dma_w(0, 0);
    f32x4v xn[8];
    float mkn;
    load_x(0, xn, mkn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) dma_w(s + 1, (s + 1) & 1);
        // columns S xo - 1 .. S xo + 4 of the 8 channels: the lane's own four, the left neighbour's last and the right neighbour's first
        float col[6][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const f32x4v v = xn[e] * mkn;
            col[1][e] = v[0]; col[2][e] = v[1]; col[3][e] = v[2]; col[4][e] = v[3];
            col[0][e] = dpp_mov<0x138, 0xf>(0.f, v[3]) * lmask;                        // wave_shr 1: lane n takes lane n - 1
            col[5][e] = dpp_mov<0x130, 0xf>(0.f, v[0]) * rmask;                        // wave_shl 1: lane n takes lane n + 1
        }
        load_x(s + 1, xn, mkn);                                                        // past the end: clamped, never used
        u32x4 xl[6][3];
#pragma unroll
        for (int c = 0; c < 6; ++c) split8(col[c], xl[c][0], xl[c][1], xl[c][2]);
        const u32x4* Wc = Ws[s & 1] + lane;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const u32x4* wp = Wc + (kx * MTW + m) * 192;
                const u32x4 wl[3] = {wp[0], wp[64], wp[128]};
#pragma unroll
                for (int j = 0; j < NPX; ++j) mac6(wl, xl[S * j + kx], acc[m][j], alo[m][j]);     // pixel j: column S (xo + j) - 1 + kx
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: out = relu?(acc + bias) + res1 + res2, rows (r & 3) + 8 (r >> 2) + 4 kh of each row block, one 4 NPX-byte access per row
    if (live) {
        typedef float fpx __attribute__((ext_vector_type(NPX)));
        const float lo = k.relu ? 0.f : -3.402823466e38f;
        const int64_t ob = (int64_t)b * k.Cout * Lo + pc;
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < k.Cout) {
                    const float bv = k.bias ? k.bias[row] : 0.f;
                    const int64_t o = ob + (int64_t)row * Lo;
                    fpx v;
#pragma unroll
                    for (int j = 0; j < NPX; ++j) v[j] = fmaxf(acc[m][j][r] + alo[m][j][r] + bv, lo);
                    if (k.res1) v += *reinterpret_cast<const fpx*>(k.res1 + o);
                    if (k.res2) v += *reinterpret_cast<const fpx*>(k.res2 + o);
                    *reinterpret_cast<fpx*>(k.out + o) = v;
                }
            }
    }
}

}  // namespace

// return 1 when the shape is one the row form takes (the callers fall back to the shifted-tap form otherwise)
extern "C" int bem_conv4x4s2_fast_supported(int Cin, int H, int W) {
    const int Wo = W / 2;
    return Cin % 8 == 0 && H % 2 == 0 && W % 2 == 0 && Wo >= 2 && Wo <= 64 && (Wo & (Wo - 1)) == 0;
}
extern "C" int bem_conv3x3_rows_supported(int Cin, int H, int W) {
    return Cin % 8 == 0 && H > 0 && W >= 4 && W <= 128 && (W & (W - 1)) == 0;
}

// KS = 4: the 4x4 stride-2 form; KS = 3: 3x3 stride 1.  Row blocks of output channels in pairs where the registers allow (4x4), singly else.
int conv_rows_launch(int KS, const float* x, int64_t x_bstride, const float* Wp, const float* bias, const float* res1, const float* res2, float* out,
                     int B, int Cin, int H, int W, int Cout, int relu, void* stream) {
    const char* what = KS == 4 ? "conv4x4s2_x6" : "conv3x3_x6";
    BEM_REQUIRE(x && Wp && out, "%s: null tensor", what);
    BEM_REQUIRE(B >= 0 && B <= 65535 && Cin > 0 && Cout > 0 && (KS == 4 ? bem_conv4x4s2_fast_supported(Cin, H, W) : bem_conv3x3_rows_supported(Cin, H, W)),
                "%s: shape outside the row form", what);
    BEM_REQUIRE((((uintptr_t)Wp | (uintptr_t)x | (uintptr_t)out | (uintptr_t)(res1 ? res1 : out) | (uintptr_t)(res2 ? res2 : out)) & 15) == 0 && (x_bstride % 4) == 0,
                "%s: alignment (x, packed weights, out and residuals 16 bytes)", what);
    const int S = KS == 4 ? 2 : 1, Ho = H / S, Wo = W / S;
    BEM_REQUIRE((int64_t)Cout * Ho * Wo < (1ll << 30) && (int64_t)Cin * H * W < (1ll << 30), "%s: plane set too large for 32-bit lane offsets", what);
    if (B == 0) return BEM_OK;
    CrX k;
    k.x = x; k.x_bs = x_bstride; k.Wp = reinterpret_cast<const u32x4*>(Wp); k.bias = bias; k.res1 = res1; k.res2 = res2; k.out = out;
    k.Cin = Cin; k.H = H; k.W = W; k.Ho = Ho; k.Wo = Wo; k.Cout = Cout; k.KB = cdiv(Cin, 16); k.MT = cdiv(Cout, 32); k.relu = relu;
    k.wo_shift = __builtin_ctz(Wo);
    hipStream_t s = (hipStream_t)stream;
    if (KS == 4) {
        const int pairs = k.MT / 2, nx = cdiv(Ho * Wo, 256);
        if (pairs) {
            k.mt_first = 0;
            conv_rows_x6_kernel<2, 4, 2><<<dim3(nx, pairs, B), 256, 0, s>>>(k);
        }
        if (k.MT & 1) {
            k.mt_first = 2 * pairs;
            conv_rows_x6_kernel<1, 4, 2><<<dim3(nx, 1, B), 256, 0, s>>>(k);
        }
    } else {
        k.mt_first = 0;
        conv_rows_x6_kernel<1, 3, 1><<<dim3(cdiv(Ho * Wo, 512), k.MT, B), 256, 0, s>>>(k);
    }
    return bem_check_launch(what);
}
