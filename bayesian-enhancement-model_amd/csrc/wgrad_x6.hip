// Weight gradient of a 1x1 layer on the bf16 matrix cores:  dW[m][c] += sum_{b,p} dy[b][m][p] * x[b][c][p]   (+ dbias = row sums of dy)
// -- the same contract as bem_pw_wgrad_f32 (wgrad.hip), for planes with L % 16 == 0.
//
// Why a second form: with pixels as the GEMM's K dimension both operands are "row = channel, K = contiguous pixels".  The f32
// instruction v_mfma_f32_32x32x2_f32 wants two K values per lane, i.e. 8-byte pieces of 32 different planes per half-wave, so
// wgrad.hip stages and transposes both operands through LDS (1.1 TB/s over the training step's launches).  The bf16 instruction
// v_mfma_f32_32x32x16_bf16 wants EIGHT consecutive K values per lane -- 32 contiguous bytes of one plane -- for A and for B alike:
// both operands are loaded straight from global memory in operand order, split into three bf16 limbs in registers (exact, see
// pw_gemm_x6.hip) and multiplied as six limb products into f32 accumulators.  No LDS on the load path, no transposes.
//
// Mapping: a wave walks over 32-pixel chunks (two MFMA K-steps), in runs of consecutive chunks strided over all waves of grid.x;
// grid.y / grid.z select a group of MTW M-tiles (dy rows) and NTW N-tiles (x rows).  Per chunk a lane loads 64 contiguous bytes of
// its dy row and of its x row per tile -- a half-wave pair covers one full 128-byte line per row -- one chunk ahead of the
// arithmetic (triple-buffered in registers; the kernel takes a SIMD's whole register file, one wave per SIMD).  The four waves
// of a workgroup walk the same chunks and own different tile groups (shared operand rows come from L1 / L2); every wave stores
// its partial tiles to the workgroup row's workspace slot and a second small kernel sums the slots into dW (16 adders per element): with float atomics the few hundred output lines were the bottleneck (130 of 380 us).
#include "bem_common.h"
#include "x6_common.h"
#include <stdlib.h>
#include "../../include/bem_hip.h"

namespace {

struct WxK {
    const float* dy; int64_t dy_bs; int M;
    const float* x1; int64_t x1_bs; int C1;
    const float* x2; int64_t x2_bs; int C2;
    float* dw; int64_t ldw; int blk; int perm[4];
    float* dbias;
    int L, K, MT, NT, cpi, nq;        // cpi = chunks per image (L / 32), nq = B * cpi
    float* ws; int64_t slot_elems;    // workspace: grid.x slots of M * K (+ M) partial sums
    int wy, wz;                       // waves of a workgroup: wy M-tile groups x wz N-tile groups (wy * wz <= 4)
};

template <int MTW, int NTW>
__global__ __launch_bounds__(256, 1) void wgrad_x6_kernel(WxK k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5, n = lane & 31;
    // the four waves of a workgroup walk the SAME pixel chunks and own different tile groups (k.wy x k.wz of them): the operand a
    // group shares with its neighbour comes out of the CU's L1 / the XCD's L2, so HBM sees dy and x about once
    // (with fewer than four tile groups the spare waves take chunk runs of their own: sub-rows of the workgroup, one slot each)
    const int ng = k.wy * k.wz, grp = wave % ng, sub = wave / ng, nsub = 4 / ng;
    const int wyi = grp % k.wy, wzi = grp / k.wy;
    const int mt0 = (blockIdx.y * k.wy + wyi) * MTW, nt0 = (blockIdx.z * k.wz + wzi) * NTW;
    if (mt0 >= k.MT || nt0 >= k.NT || sub >= nsub) return;       // no barrier below: idle waves may leave

    // K mapping inside a 32-pixel chunk: lanes 0-31 (kh = 0) own pixels 0..15 of their row, lanes 32-63 pixels 16..31 -- one full
    // 128-byte line per row and load group.  MFMA 0 of the chunk multiplies pixels {0..7, 16..23}, MFMA 1 pixels {8..15, 24..31};
    // A and B use the same mapping, so the sum over K is the sum over all 32 pixels.
    const float* arow[MTW];
    float am[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const int row = (mt0 + m) * 32 + n;
        arow[m] = k.dy + (int64_t)min(row, k.M - 1) * k.L + 16 * kh;
        am[m] = row < k.M ? 1.f : 0.f;
    }
    const float* brow[NTW];
    int64_t bbs[NTW];
    float bm[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int c = (nt0 + t) * 32 + n;
        const int cc = min(c, k.K - 1);
        const bool first = cc < k.C1;
        brow[t] = (first ? k.x1 + (int64_t)cc * k.L : k.x2 + (int64_t)(cc - k.C1) * k.L) + 16 * kh;
        bbs[t] = first ? k.x1_bs : k.x2_bs;
        bm[t] = c < k.K ? 1.f : 0.f;
    }
    f32x16 acc[MTW][NTW], alo[MTW][NTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = alo[m][t][r] = 0.f;
    float rs[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) rs[m] = 0.f;

    // a wave takes runs of RUN consecutive chunks (RUN * 128 bytes of every row), runs strided over all waves of grid.x; one chunk
    // (64 registers of raw operands) is always in flight under the arithmetic of the current one
    constexpr int RUN = 4;
    const int nwave = gridDim.x * nsub, w0 = blockIdx.x * nsub + sub;
    auto load = [&](int qq, float4 (&ra)[MTW][4], float4 (&rb)[NTW][4]) {
        const int b = qq / k.cpi, p0 = (qq - b * k.cpi) * 32;
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const float4* p = reinterpret_cast<const float4*>(arow[m] + (int64_t)b * k.dy_bs + p0);
#pragma unroll
            for (int u = 0; u < 4; ++u) ra[m][u] = p[u];
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float4* p = reinterpret_cast<const float4*>(brow[t] + (int64_t)b * bbs[t] + p0);
#pragma unroll
            for (int u = 0; u < 4; ++u) rb[t][u] = p[u];
        }
    };
    auto next_q = [&](int q) {                                   // successor of chunk q in this wave's order (or -1)
        int nq1 = q + 1;
        if (nq1 % RUN == 0) nq1 += (nwave - 1) * RUN;
        return nq1 < k.nq ? nq1 : -1;
    };
    auto compute = [&](const float4 (&ra)[MTW][4], const float4 (&rb)[NTW][4]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 al[MTW][3], xl[NTW][3];
#pragma unroll
            for (int m = 0; m < MTW; ++m) {
                const float4 a0 = ra[m][2 * h], a1 = ra[m][2 * h + 1];
                const float v[8] = {a0.x * am[m], a0.y * am[m], a0.z * am[m], a0.w * am[m], a1.x * am[m], a1.y * am[m], a1.z * am[m], a1.w * am[m]};
                rs[m] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                split8(v, al[m][0], al[m][1], al[m][2]);
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float4 b0 = rb[t][2 * h], b1 = rb[t][2 * h + 1];
                const float v[8] = {b0.x * bm[t], b0.y * bm[t], b0.z * bm[t], b0.w * bm[t], b1.x * bm[t], b1.y * bm[t], b1.z * bm[t], b1.w * bm[t]};
                split8(v, xl[t][0], xl[t][1], xl[t][2]);
            }
#pragma unroll
            for (int m = 0; m < MTW; ++m)
#pragma unroll
                for (int t = 0; t < NTW; ++t) mac6(al[m], xl[t], acc[m][t], alo[m][t]);
        }
    };
    // three rotating register buffers: two chunks (2 x 16 KB per wave) are in flight under the arithmetic of the current one
    float4 ra0[MTW][4], rb0[NTW][4], ra1[MTW][4], rb1[NTW][4], ra2[MTW][4], rb2[NTW][4];
    int q = w0 * RUN < k.nq ? w0 * RUN : -1;
    int q1 = q >= 0 ? next_q(q) : -1;
    if (q >= 0) load(q, ra0, rb0);
    if (q >= 0) load(q1 >= 0 ? q1 : q, ra1, rb1);
    while (q >= 0) {                                             // three chunks per trip: the buffers rotate without register copies
        int q2 = q1 >= 0 ? next_q(q1) : -1;
        load(q2 >= 0 ? q2 : q, ra2, rb2);
        compute(ra0, rb0);
        if (q1 < 0) break;
        int q3 = q2 >= 0 ? next_q(q2) : -1;
        load(q3 >= 0 ? q3 : q1, ra0, rb0);
        compute(ra1, rb1);
        if (q2 < 0) break;
        int q4 = q3 >= 0 ? next_q(q3) : -1;
        load(q4 >= 0 ? q4 : q2, ra1, rb1);
        compute(ra2, rb2);
        q = q3; q1 = q4;
    }
    // this wave's partial tiles go to the workgroup row's slot of the workspace (plain stores: 128 contiguous bytes per accumulator
    // row); wgrad_x6_reduce_kernel sums the slots into dW
    float* slot = k.ws + (int64_t)w0 * k.slot_elems;
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int gc = (nt0 + t) * 32 + n;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (gm < k.M && gc < k.K) slot[(int64_t)gm * k.K + gc] = acc[m][t][r] + alo[m][t][r];
            }
        }
    if (k.dbias && nt0 == 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const float sm = rs[m] + __shfl_xor(rs[m], 32, 64);
            const int gm = (mt0 + m) * 32 + n;
            if (kh == 0 && gm < k.M) slot[(int64_t)k.M * k.K + gm] = sm;
        }
    }
}

// dW[row map(m)][c] += sum over the gx slots; dbias likewise.  grid (elements / 256, slot groups): a thread sums its group's slots
// (reads coalesced across threads, four independent partial sums) and adds the result with one atomic -- gridDim.y adders per element.
__global__ void wgrad_x6_reduce_kernel(WxK k, int gx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nmk = (int64_t)k.M * k.K, tot = nmk + (k.dbias ? k.M : 0);
    if (i >= tot) return;
    const int per = (gx + gridDim.y - 1) / gridDim.y, g0 = blockIdx.y * per, g1 = min(gx, g0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* p = k.ws + i;
    int g = g0;
    for (; g + 3 < g1; g += 4) {
        s0 += p[(int64_t)g * k.slot_elems]; s1 += p[(int64_t)(g + 1) * k.slot_elems];
        s2 += p[(int64_t)(g + 2) * k.slot_elems]; s3 += p[(int64_t)(g + 3) * k.slot_elems];
    }
    for (; g < g1; ++g) s0 += p[(int64_t)g * k.slot_elems];
    const float s = (s0 + s1) + (s2 + s3);
    if (g0 >= g1) return;
    if (i < nmk) {
        const int gm = (int)(i / k.K), gc = (int)(i - (int64_t)gm * k.K);
        const int blkid = gm / k.blk;
        const int orow = k.perm[blkid] * k.blk + (gm - blkid * k.blk);
        atomicAdd(k.dw + (int64_t)orow * k.ldw + gc, s);
    } else {
        atomicAdd(k.dbias + (i - nmk), s);
    }
}

}  // namespace

struct WgxGrid { int gx, gy, gz, wy, wz, nslot; };
static WgxGrid wgx_grid(int M, int K, int B, int L) {
    const int py = cdiv(cdiv(M, 32), 2), pz = cdiv(cdiv(K, 32), 2);          // tile groups (2 x 2 tiles each) along M and along K
    WgxGrid g;
    g.wy = std::min(4, py);
    g.wz = std::min(4 / g.wy, pz);
    g.gy = cdiv(py, g.wy); g.gz = cdiv(pz, g.wz);
    const int64_t nq = (int64_t)B * (L / 32);
    // one workgroup per CU (a wave takes a SIMD's whole register file): ~512 workgroups in all, each at least two runs of 4 chunks
    g.gx = std::max(1, 512 / (g.gy * g.gz));
    const int nsub = 4 / (g.wy * g.wz);
    g.gx = (int)std::min<int64_t>(g.gx, std::max<int64_t>(1, nq / (8 * nsub)));
    g.nslot = g.gx * nsub;
    return g;
}

extern "C" int64_t bem_pw_wgrad_x6_ws_elems(int M, int K, int B, int L) {
    if (M <= 0 || K <= 0 || B <= 0 || L <= 0) return 0;
    return (int64_t)wgx_grid(M, K, B, L).nslot * ((int64_t)M * K + M);
}

extern "C" int bem_pw_wgrad_x6_f32(const bem_wgrad_args* a, float* ws, int64_t ws_elems, void* stream) {
    BEM_REQUIRE(a && a->dy && a->x1 && a->dw && ws, "pw_wgrad_x6: null pointer");
    BEM_REQUIRE(a->B > 0 && a->M > 0 && a->L > 0 && a->C1 > 0 && a->C2 >= 0, "pw_wgrad_x6: bad sizes");
    BEM_REQUIRE(a->C2 == 0 || a->x2, "pw_wgrad_x6: C2 > 0 needs x2");
    BEM_REQUIRE(a->L % 32 == 0, "pw_wgrad_x6: L = %d is not a multiple of 32 (use bem_pw_wgrad_f32)", a->L);
    WxK k{};
    k.K = a->C1 + a->C2;
    BEM_REQUIRE(a->ldw >= k.K, "pw_wgrad_x6: ldw < K");
    k.blk = a->blk_rows > 0 ? a->blk_rows : a->M;
    BEM_REQUIRE(a->M % k.blk == 0 && a->M / k.blk <= 4, "pw_wgrad_x6: at most 4 row blocks");
    for (int i = 0; i < 4; ++i) {
        k.perm[i] = a->blk_rows > 0 ? a->perm[i] : i;
        BEM_REQUIRE(k.perm[i] >= 0 && k.perm[i] < 4, "pw_wgrad_x6: bad row-block permutation");
    }
    k.dy = a->dy; k.dy_bs = a->dy_bstride ? a->dy_bstride : (int64_t)a->M * a->L; k.M = a->M;
    k.x1 = a->x1; k.x1_bs = a->x1_bstride ? a->x1_bstride : (int64_t)a->C1 * a->L; k.C1 = a->C1;
    k.x2 = a->x2 ? a->x2 : a->x1; k.x2_bs = a->x2_bstride ? a->x2_bstride : (int64_t)a->C2 * a->L; k.C2 = a->C2;
    BEM_REQUIRE((k.dy_bs & 3) == 0 && (k.x1_bs & 3) == 0 && (k.x2_bs & 3) == 0 && ((uintptr_t)k.dy & 15) == 0 && ((uintptr_t)k.x1 & 15) == 0 &&
                ((uintptr_t)k.x2 & 15) == 0, "pw_wgrad_x6: 16-byte aligned operands / strides expected");
    k.dw = a->dw; k.ldw = a->ldw; k.dbias = a->dbias;
    k.L = a->L; k.MT = cdiv(a->M, 32); k.NT = cdiv(k.K, 32); k.cpi = a->L / 32;
    const int64_t nq = (int64_t)a->B * k.cpi;
    BEM_REQUIRE(nq < (1ll << 31), "pw_wgrad_x6: too many pixel chunks");
    k.nq = (int)nq;
    const WgxGrid gg = wgx_grid(a->M, k.K, a->B, a->L);
    const int gx = gg.nslot;          // slots to sum
    k.wy = gg.wy; k.wz = gg.wz;
    k.ws = ws; k.slot_elems = (int64_t)a->M * k.K + a->M;
    BEM_REQUIRE(ws_elems >= (int64_t)gx * k.slot_elems, "pw_wgrad_x6: workspace of %lld floats, bem_pw_wgrad_x6_ws_elems asks for %lld",
                (long long)ws_elems, (long long)((int64_t)gx * k.slot_elems));
    hipStream_t st = (hipStream_t)stream;
    wgrad_x6_kernel<2, 2><<<dim3(gg.gx, gg.gy, gg.gz), 256, 0, st>>>(k);
    const int64_t tot = (int64_t)a->M * k.K + (a->dbias ? a->M : 0);
    wgrad_x6_reduce_kernel<<<dim3((unsigned)cdiv64(tot, 256), std::min(gx, 16)), 256, 0, st>>>(k, gx);
    return bem_check_launch("pw_wgrad_x6");
}
