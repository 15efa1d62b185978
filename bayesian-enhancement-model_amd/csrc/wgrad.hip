// Weight-gradient GEMM of the training step: dW[m][n] += sum_{b,p} A[b][m][p] * Bop[b][n][p]   ("NT over pixels").
//
// In channel-planar NCHW both operands have the contraction index (the pixel) contiguous, so this is the one GEMM of the
// path whose K dimension is the long, coalesced one: A = dL/dout rows (M = Cout), Bop = the layer's input rows (N = Cin for a
// 1x1 layer, or Cin*KH*KW rows gathered on the fly for a dense KHxKW convolution -- no im2col tensor in HBM).
// Replaces what autograd runs for the reference's nn.Conv2d / Linear2d weights (vmamba.py:42-55,123-125; arch convs
// DecompDualBranchDDWavelet_arch.py:40-51,190,217-233) inside image_enhancer_model.py:200 (`backward()`).
//
// Kernel: 256 threads = 4 wavefronts; a workgroup owns an (MC = 128*MTW) x (NC = 32*NTW) block of dW and a contiguous range of
// 32-pixel tiles.  Per tile the MC + NC row segments (128 B each, whole cache lines) go global -> registers -> LDS (row
// stride 33 floats: conflict-free operand reads), the next tile's loads are in flight while the matrix cores run
// v_mfma_f32_32x32x2_f32 over the 16 pixel pairs of the tile (A[i][k]: lane (i = l & 31, k = l >> 5) -- exactly one LDS dword).
// Each wave owns MTW M-tiles x all NTW N-tiles (16 accumulator registers each).  Partial blocks are added to dW with float
// atomics (two 128-byte row segments per wave-instruction: the full-rate shape).
#include "bem_common.h"
#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgK {
    const float* a; int64_t a_bs;
    const float* b1; int64_t b1_bs; int C1;
    const float* b2; int64_t b2_bs; int C2;
    int conv, KH, KW, S, PAD, Hin, Win, Wout;
    float* out; int64_t ldo; int blk_rows; int perm[4];
    float* rowsum;
    int B, M, N, L;
    int tiles;            // ceil(L / 32)
    int chunks_per_wg;    // (b, tile) pairs per workgroup
    int total_chunks;
};

constexpr int PT = 32, LDR = 33;

template <int MTW, int NTW, bool CONV, bool VEC>      // VEC: L % 4 == 0 (a compile-time variant: a runtime flag puts a branch next to every load)
__global__ __launch_bounds__(256, (MTW * NTW >= 6 ? 1 : 2)) void wgrad_kernel(WgK k) {
    constexpr int MC = 128 * MTW, NC = 32 * NTW, NP = 4 * MTW + NTW;     // NP passes of 32 rows
    extern __shared__ float lds[];                                       // [(MC + NC)][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * MC, n0 = blockIdx.z * NC;
    const int c_begin = blockIdx.x * k.chunks_per_wg;
    const int c_end = min(c_begin + k.chunks_per_wg, k.total_chunks);
    constexpr bool vec = VEC;
    const int lrow = tid >> 3, lq = tid & 7;                             // staging role: row within a 32-row pass, float4 slot

    f32x16 acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    float rs[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) rs[i] = 0.f;

    // Addressing: one wave-uniform base pointer per operand and tile (SGPRs) + a 32-bit element offset per staging row; rows outside
    // M / N read a clamped row and are multiplied by 0 when the tile is written to LDS (no branch next to a load: the NP loads
    // of a tile issue back to back).  Per-row pointers held in registers spilled (14 x 64 bit on top of 96 accumulators).
    // CONV: (ci, ky, kx) of this thread's gathered rows
    int cci[NTW], cky[NTW], ckx[NTW];
    if (CONV) {
        const int T = k.KH * k.KW;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = min(n0 + j * 32 + lrow, k.N - 1);
            cci[j] = n / T;
            const int tap = n - cci[j] * T;
            cky[j] = tap / k.KW;
            ckx[j] = tap - cky[j] * k.KW;
        }
    }

    float4 stage[NP];
    auto fetch = [&](int chunk) {
        const int b = chunk / k.tiles, tile = chunk - b * k.tiles;
        const int p = tile * PT + 4 * lq;
        const int pc = vec ? min(p, k.L - 4) : 0;
        const float* abase = k.a + (int64_t)b * k.a_bs;
        const float* b1base = k.b1 + (int64_t)b * k.b1_bs;
        const float* b2base = CONV ? b1base : k.b2 + (int64_t)b * k.b2_bs - (int64_t)k.C1 * k.L;   // indexed by the concatenated row
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            float4 v;
            if (CONV && u >= 4 * MTW) {
                const int j = u - 4 * MTW;
                const float* src = b1base + (int64_t)cci[j] * k.Hin * k.Win;
                float e[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int pp = min(p + t, k.L - 1);
                    const int oi = pp / k.Wout, oj = pp - oi * k.Wout;
                    const int y = oi * k.S + cky[j] - k.PAD, x = oj * k.S + ckx[j] - k.PAD;
                    const bool ok = y >= 0 && y < k.Hin && x >= 0 && x < k.Win;
                    e[t] = src[min(max(y, 0), k.Hin - 1) * k.Win + min(max(x, 0), k.Win - 1)] * (ok ? 1.f : 0.f);
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            } else {
                const float* base;
                unsigned off;
                if (u < 4 * MTW) {
                    base = abase;
                    off = (unsigned)min(m0 + u * 32 + lrow, k.M - 1) * (unsigned)k.L;
                } else {
                    const int n = min(n0 + (u - 4 * MTW) * 32 + lrow, k.N - 1);
                    base = n < k.C1 ? b1base : b2base;
                    off = (unsigned)n * (unsigned)k.L;
                }
                if (vec) {
                    v = *reinterpret_cast<const float4*>(base + off + pc);
                } else {
                    v = make_float4(base[off + min(p, k.L - 1)], base[off + min(p + 1, k.L - 1)], base[off + min(p + 2, k.L - 1)], base[off + min(p + 3, k.L - 1)]);
                }
            }
            stage[u] = v;          // raw: the masks are applied when the tile is written to LDS, so nothing here waits for the loads
        }
    };

    if (c_begin < c_end) fetch(c_begin);
    for (int chunk = c_begin; chunk < c_end; ++chunk) {
        __syncthreads();                       // the previous tile's operand reads are done
        {
            const int p = (chunk % k.tiles) * PT + 4 * lq;
            const float m0v = p < k.L ? 1.f : 0.f, m1v = p + 1 < k.L ? 1.f : 0.f, m2v = p + 2 < k.L ? 1.f : 0.f, m3v = p + 3 < k.L ? 1.f : 0.f;
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                float* d = lds + (u * 32 + lrow) * LDR + 4 * lq;
                const float rm = (u < 4 * MTW ? m0 + u * 32 + lrow < k.M : n0 + (u - 4 * MTW) * 32 + lrow < k.N) ? 1.f : 0.f;
                d[0] = stage[u].x * (rm * m0v); d[1] = stage[u].y * (rm * m1v); d[2] = stage[u].z * (rm * m2v); d[3] = stage[u].w * (rm * m3v);
            }
        }
        __syncthreads();
        if (chunk + 1 < c_end) fetch(chunk + 1);
        const float* As = lds + (wave * MTW * 32 + r) * LDR + h;
        const float* Bs = lds + (MC + r) * LDR + h;
#pragma unroll
        for (int s = 0; s < PT / 2; ++s) {
            float av[MTW], bv[NTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) av[i] = As[i * 32 * LDR + 2 * s];
#pragma unroll
            for (int j = 0; j < NTW; ++j) bv[j] = Bs[j * 32 * LDR + 2 * s];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                rs[i] += av[i];
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: D[row = (q & 3) + 8 (q >> 2) + 4 h][col = r]
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int mb = m0 + (wave * MTW + i) * 32;
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = n0 + j * 32 + r;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = mb + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m < k.M && n < k.N) {
                    const int mo = k.perm[m / k.blk_rows] * k.blk_rows + m % k.blk_rows;
                    atomicAdd(k.out + (int64_t)mo * k.ldo + n, acc[i][j][q]);
                }
            }
        }
        if (k.rowsum && blockIdx.z == 0) {
            const float t = rs[i] + __shfl_xor(rs[i], 32, BEM_WAVE);
            const int m = mb + r;
            if (h == 0 && m < k.M) atomicAdd(k.rowsum + m, t);
        }
    }
}

template <int MTW, int NTW, bool CONV, bool VEC>
int launch_wgrad_t(WgK k, hipStream_t s) {
    constexpr int MC = 128 * MTW, NC = 32 * NTW;
    const int ny = cdiv(k.M, MC), nz = cdiv(k.N, NC);
    k.tiles = cdiv(k.L, PT);
    k.total_chunks = k.B * k.tiles;
    // every pixel split ends in MC x NC float atomics onto the same dW block: few, long splits (about two workgroups per CU)
    const int target = std::max(1, 512 / (ny * nz));
    // ... unless the whole problem is a handful of chunks (Stage I: 8x8 .. 2x2 planes): there one workgroup walking them one after the
    // other is pure latency, and the few extra atomics cost nothing
    k.chunks_per_wg = std::max(k.total_chunks >= 256 ? 8 : (k.total_chunks >= 32 ? 2 : 1), cdiv(k.total_chunks, target));
    const int nx = cdiv(k.total_chunks, k.chunks_per_wg);
    const size_t shm = (size_t)(MC + NC) * LDR * sizeof(float);
    static_assert((MC + NC) * LDR * sizeof(float) <= 64 * 1024, "dynamic LDS above 64 KiB needs hipFuncSetAttribute");
    wgrad_kernel<MTW, NTW, CONV, VEC><<<dim3(nx, ny, nz), 256, shm, s>>>(k);
    return bem_check_launch("wgrad");
}
template <int MTW, int NTW>
int launch_wgrad(const WgK& k, hipStream_t s) {
    return k.conv ? launch_wgrad_t<MTW, NTW, true, true>(k, s) : launch_wgrad_t<MTW, NTW, false, true>(k, s);
}

int dispatch_wgrad(const WgK& k, hipStream_t s) {
    if (k.L & 3)          // ragged planes (tests, odd crops): one scalar-load variant
        return k.conv ? launch_wgrad_t<1, 2, true, false>(k, s) : launch_wgrad_t<1, 2, false, false>(k, s);
    const int mt = cdiv(k.M, 32), nt = cdiv(k.N, 32);
    const int ntw = nt >= 4 ? 5 : nt;                       // 1, 2, 3 or 5 N-tiles per workgroup
    int mtw = ntw == 5 ? 1 : (ntw == 3 ? 2 : 3);            // MTW * NTW <= 6 accumulator tiles per wave
    mtw = std::min(mtw, std::max(1, cdiv(mt, 4)));
    switch (ntw * 10 + mtw) {
        case 11: return launch_wgrad<1, 1>(k, s);
        case 12: return launch_wgrad<2, 1>(k, s);
        case 13: return launch_wgrad<3, 1>(k, s);
        case 21: return launch_wgrad<1, 2>(k, s);
        case 22: return launch_wgrad<2, 2>(k, s);
        case 23: return launch_wgrad<3, 2>(k, s);
        case 31: return launch_wgrad<1, 3>(k, s);
        case 32: return launch_wgrad<2, 3>(k, s);
        default: return launch_wgrad<1, 5>(k, s);
    }
}

}  // namespace

extern "C" int bem_pw_wgrad_f32(const bem_wgrad_args* a, void* stream) {
    BEM_REQUIRE(a && a->dy && a->x1 && a->dw, "pw_wgrad: null pointer");
    BEM_REQUIRE(a->B > 0 && a->M > 0 && a->L > 0 && a->C1 > 0 && a->C2 >= 0, "pw_wgrad: bad sizes");
    BEM_REQUIRE(a->C2 == 0 || a->x2, "pw_wgrad: C2 > 0 needs x2");
    const int N = a->C1 + a->C2;
    BEM_REQUIRE(a->ldw >= N, "pw_wgrad: ldw < K");
    const int blk = a->blk_rows > 0 ? a->blk_rows : a->M;
    BEM_REQUIRE(a->M % blk == 0 && a->M / blk <= 4, "pw_wgrad: at most 4 row blocks");
    WgK k{};
    k.a = a->dy; k.a_bs = a->dy_bstride ? a->dy_bstride : (int64_t)a->M * a->L;
    k.b1 = a->x1; k.b1_bs = a->x1_bstride ? a->x1_bstride : (int64_t)a->C1 * a->L; k.C1 = a->C1;
    k.b2 = a->x2; k.b2_bs = a->x2_bstride ? a->x2_bstride : (int64_t)a->C2 * a->L; k.C2 = a->C2;
    k.conv = 0;
    k.out = a->dw; k.ldo = a->ldw; k.blk_rows = blk;
    for (int i = 0; i < 4; ++i) {
        k.perm[i] = a->blk_rows > 0 ? a->perm[i] : i;
        BEM_REQUIRE(k.perm[i] >= 0 && k.perm[i] < 4, "pw_wgrad: bad row-block permutation");
    }
    k.rowsum = a->dbias;
    k.B = a->B; k.M = a->M; k.N = N; k.L = a->L;
    if (((k.L & 3) == 0)) {
        BEM_REQUIRE((k.a_bs & 3) == 0 && (k.b1_bs & 3) == 0 && (k.b2_bs & 3) == 0 && ((uintptr_t)k.a & 15) == 0 && ((uintptr_t)k.b1 & 15) == 0 &&
                    ((uintptr_t)k.b2 & 15) == 0, "pw_wgrad: 16-byte aligned operands / strides expected when L %% 4 == 0");
    }
    return dispatch_wgrad(k, (hipStream_t)stream);
}

extern "C" int bem_conv_wgrad_f32(const float* dy, const float* x, int64_t x_bstride, float* dw, float* dbias, int B, int Cin, int H, int W,
                                  int Cout, int KH, int KW, int stride, int pad, void* stream) {
    BEM_REQUIRE(dy && x && dw, "conv_wgrad: null pointer");
    BEM_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "conv_wgrad: bad sizes");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    BEM_REQUIRE(Ho > 0 && Wo > 0, "conv_wgrad: empty output");
    WgK k{};
    k.a = dy; k.a_bs = (int64_t)Cout * Ho * Wo;
    k.b1 = x; k.b1_bs = x_bstride ? x_bstride : (int64_t)Cin * H * W; k.C1 = Cin * KH * KW;
    k.b2 = nullptr; k.b2_bs = 0; k.C2 = 0;
    k.conv = 1; k.KH = KH; k.KW = KW; k.S = stride; k.PAD = pad; k.Hin = H; k.Win = W; k.Wout = Wo;
    k.out = dw; k.ldo = (int64_t)Cin * KH * KW; k.blk_rows = Cout;
    for (int i = 0; i < 4; ++i) k.perm[i] = i;
    k.rowsum = dbias;
    k.B = B; k.M = Cout; k.N = Cin * KH * KW; k.L = Ho * Wo;
    return dispatch_wgrad(k, (hipStream_t)stream);
}
