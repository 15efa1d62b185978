// Backward of the fused SS2D core (bem_ss2d_scan_f32): what autograd runs through
//   cross_scan -> x_proj split -> dt_proj -> softplus -> selective scan (4 directions, N = 1) -> cross_merge
// in the reference (vmamba.py:657-684; CrossScanF / CrossMergeF.backward csm_triton.py:207-273; SelectiveScanCuda.backward
// csms6s.py:95-113 -> selective_scan_bwd_kernel_oflex.cuh:73-289), without materialising the 4x expanded tensors.
//
// One workgroup per (channel c, image b, orientation o), like the forward's general form: it handles directions k = o (scan
// order = memory order) and k = o + 2 (reverse).  Per direction:
//   pass 1  recompute the forward recurrence chunk by chunk, keeping the state entering each chunk (LDS);
//   pass 2  chunks in reverse scan order: rebuild h_t, run the adjoint recurrence dh_t = C_t dy_t + a_next dh_next with the
//           mirrored block scan (a_next of a thread's last element comes from its neighbour by a lane shift / LDS), then
//             dx_t   += D dy_t + dh_t dt_t B_t                       (row of this workgroup: plain stores)
//             ddt_t   = dh_t (B_t x_t + A a_t h_prev) sigmoid(z_t)    z = dt_proj row + bias (softplus threshold 20)
//             dxd[r][t] += ddt_t wdt[r], dxd[R][t] += dh_t dt_t x_t, dxd[R+1][t] += dy_t h_t      (float atomics: shared by the
//                          C channels of the image, like the reference's dB / dC)
//             dA += sum dh dt a h_prev, dD += sum dy x, ddtb += sum ddt, ddtw[r] += sum ddt xd[r]   (block sums, one atomic each)
// dA is returned as the gradient of A_logs (A = -exp(A_logs): dA_logs = dA * A).
#include "scan_common.h"

namespace {

constexpr int SB_MAXCH = 128;

template <int NT, int E, bool REV>
__device__ __forceinline__ void ss2d_dir_bwd(const float* __restrict__ xr, const float* __restrict__ dyr, const float* __restrict__ xd,
                                             float* __restrict__ dxd, float* __restrict__ dxr, const float* __restrict__ wdt, float dtb,
                                             float Ak, float Dk, int L, int R, bool first, float* agg, float* red, float* cs, float* nb,
                                             float* accw, float* dAlog_p, float* dDs_p, float* ddtb_p, float* ddtw_p) {
    constexpr int CH = NT * E, NW = NT / BEM_WAVE;
    const int nchunks = (L + CH - 1) / CH;
    const bool vec = (L % 4 == 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- pass 1: state entering every chunk, in scan order ----
    float carry = 0.f;
    for (int jj = 0; jj < nchunks; ++jj) {
        const int j = REV ? nchunks - 1 - jj : jj;
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        if (threadIdx.x == 0) cs[j] = carry;
        float x[E], a[E], bb[E], cv[E], h[E];
        load_row<E>(xr, t0, L, vec, x);
        float dts[E];
#pragma unroll
        for (int e = 0; e < E; ++e) dts[e] = 0.f;
        for (int r = 0; r < R; ++r) {
            float v[E];
            load_row<E>(xd + (int64_t)r * L, t0, L, vec, v);
            const float w = wdt[r];
#pragma unroll
            for (int e = 0; e < E; ++e) dts[e] = fmaf(w, v[e], dts[e]);
        }
        float Bv[E];
        load_row<E>(xd + (int64_t)R * L, t0, L, vec, Bv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = t0 + e < L;
            const float dl = bem_softplus(dts[e] + dtb);
            a[e] = ok ? bem_fexp(dl * Ak) : 1.f;
            bb[e] = ok ? dl * Bv[e] * x[e] : 0.f;
            cv[e] = 0.f;
        }
        block_scan_affine<NT, E, REV>(a, bb, h, carry, agg);
        (void)cv;
    }
    if (threadIdx.x < 16) accw[threadIdx.x] = 0.f;
    if (threadIdx.x == 0) nb[NW + 1] = 1.f;          // a of the first element (scan order) of the chunk processed before: none yet
    __syncthreads();
    // ---- pass 2 ----
    float cr = 0.f, accA = 0.f, accD = 0.f, accB = 0.f;
    for (int jj = nchunks - 1; jj >= 0; --jj) {
        const int j = REV ? nchunks - 1 - jj : jj;
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        float x[E], dy[E], dts[E], dl[E], a[E], bb[E], Bv[E], Cv[E], h[E], ar[E], br[E], dh[E], dz[E];
        load_row<E>(xr, t0, L, vec, x);
        load_row<E>(dyr, t0, L, vec, dy);
#pragma unroll
        for (int e = 0; e < E; ++e) dts[e] = 0.f;
        for (int r = 0; r < R; ++r) {
            float v[E];
            load_row<E>(xd + (int64_t)r * L, t0, L, vec, v);
            const float w = wdt[r];
#pragma unroll
            for (int e = 0; e < E; ++e) dts[e] = fmaf(w, v[e], dts[e]);
        }
        load_row<E>(xd + (int64_t)R * L, t0, L, vec, Bv);
        load_row<E>(xd + (int64_t)(R + 1) * L, t0, L, vec, Cv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = t0 + e < L;
            dl[e] = bem_softplus(dts[e] + dtb);
            a[e] = ok ? bem_fexp(dl[e] * Ak) : 1.f;
            bb[e] = ok ? dl[e] * Bv[e] * x[e] : 0.f;
        }
        float cf = cs[j];
        block_scan_affine<NT, E, REV>(a, bb, h, cf, agg);
        // a of the successor (scan order) of each element
        const float a_first = REV ? a[E - 1] : a[0];                 // this thread's first element in scan order
        float a_nb = REV ? __shfl_up(a_first, 1, BEM_WAVE) : __shfl_down(a_first, 1, BEM_WAVE);
        const float a_prev_chunk = nb[NW + 1];
        __syncthreads();                                             // everyone has read nb[NW + 1] of the previous iteration
        if (lane == (REV ? 63 : 0)) nb[wave] = a_first;              // first lane (scan order) of each wave
        __syncthreads();
        {
            const int edge_lane = REV ? 0 : 63;                       // last lane (scan order) of the wave
            const int wnext = REV ? wave - 1 : wave + 1;
            const bool has_next_wave = REV ? (wave > 0) : (wave < NW - 1);
            if (lane == edge_lane) a_nb = has_next_wave ? nb[wnext] : a_prev_chunk;
        }
        __syncthreads();
        if (threadIdx.x == (REV ? NT - 1 : 0)) nb[NW + 1] = a_first; // first thread (scan order) of this chunk, for the next iteration
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = t0 + e < L;
            if (REV) ar[e] = (e > 0) ? a[e - 1] : a_nb;
            else ar[e] = (e + 1 < E) ? a[e + 1] : a_nb;
            br[e] = ok ? Cv[e] * dy[e] : 0.f;
        }
        block_scan_affine<NT, E, !REV>(ar, br, dh, cr, agg);
        float dxv[E], dBv[E], dCv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = t0 + e < L;
            const float hm = h[e] - bb[e];
            const float dhd = dh[e] * dl[e];
            dxv[e] = fmaf(dhd, Bv[e], Dk * dy[e]);
            const float ddl = dh[e] * fmaf(Bv[e], x[e], Ak * hm);
            const float z = dts[e] + dtb;
            const float sg = z <= 20.f ? 1.f / (1.f + bem_fexp(-z)) : 1.f;
            dz[e] = ok ? ddl * sg : 0.f;
            dBv[e] = dhd * x[e];
            dCv[e] = dy[e] * h[e];
            if (ok) {
                accA = fmaf(dhd, hm, accA);
                accD = fmaf(dy[e], x[e], accD);
                accB += dz[e];
            }
        }
        if (!first) {
            float prev[E];
            load_row<E>(dxr, t0, L, vec, prev);
#pragma unroll
            for (int e = 0; e < E; ++e) dxv[e] += prev[e];
        }
        store_row<E>(dxr, t0, L, vec, dxv);
        for (int r = 0; r < R; ++r) {
            float v[E];
            load_row<E>(xd + (int64_t)r * L, t0, L, vec, v);
            const float w = wdt[r];
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                s = fmaf(dz[e], v[e], s);
                if (t0 + e < L) atomicAdd(dxd + (int64_t)r * L + t0 + e, dz[e] * w);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, BEM_WAVE);
            if (lane == 0) atomicAdd(&accw[r], s);
        }
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (t0 + e < L) {
                atomicAdd(dxd + (int64_t)R * L + t0 + e, dBv[e]);
                atomicAdd(dxd + (int64_t)(R + 1) * L + t0 + e, dCv[e]);
            }
    }
    accA = block_reduce_sum<NT>(accA, red);
    accD = block_reduce_sum<NT>(accD, red);
    accB = block_reduce_sum<NT>(accB, red);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dAlog_p, accA * Ak);
        atomicAdd(dDs_p, accD);
        atomicAdd(ddtb_p, accB);
    }
    if (threadIdx.x < R) atomicAdd(ddtw_p + threadIdx.x, accw[threadIdx.x]);
    __syncthreads();
}

template <int NT, int E>
__global__ __launch_bounds__(NT) void ss2d_scan_bwd_kernel(
    const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ xd0, const float* __restrict__ xd1,
    const float* __restrict__ dy0, const float* __restrict__ dy1, const float* __restrict__ dtw, const float* __restrict__ dtb,
    const float* __restrict__ A, const float* __restrict__ Ds, float* __restrict__ dx0, float* __restrict__ dx1,
    float* __restrict__ dxd0, float* __restrict__ dxd1, float* __restrict__ dAlog, float* __restrict__ dDs, float* __restrict__ ddtw,
    float* __restrict__ ddtb, int Bn, int C, int L, int R, int64_t xbs0, int64_t xbs1) {
    constexpr int NW = NT / BEM_WAVE;
    __shared__ float agg[2 * NW];
    __shared__ float red[NW];
    __shared__ float cs[SB_MAXCH];
    __shared__ float nb[NW + 2];
    __shared__ float accw[16];
    const int total = gridDim.x, lin = blockIdx.x;
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int c = wi % C, b = (wi / C) % Bn, o = wi / (C * Bn);
    const int64_t row = ((int64_t)b * C + c) * L;
    const float* xr = (o ? x1 : x0) + row;
    const float* dyr = (o ? dy1 : dy0) + row;
    float* dxr = (o ? dx1 : dx0) + row;
    const float* xd = o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0;
    float* dxd = (o ? dxd1 : dxd0) + (int64_t)b * 2 * (R + 2) * L;
    const int kf = o, kr = o + 2;
    ss2d_dir_bwd<NT, E, false>(xr, dyr, xd, dxd, dxr, dtw + ((int64_t)kf * C + c) * R, dtb[kf * C + c], A[kf * C + c], Ds[kf * C + c], L, R,
                               true, agg, red, cs, nb, accw, dAlog + kf * C + c, dDs + kf * C + c, ddtb + kf * C + c,
                               ddtw + ((int64_t)kf * C + c) * R);
    ss2d_dir_bwd<NT, E, true>(xr, dyr, xd + (int64_t)(R + 2) * L, dxd + (int64_t)(R + 2) * L, dxr, dtw + ((int64_t)kr * C + c) * R,
                              dtb[kr * C + c], A[kr * C + c], Ds[kr * C + c], L, R, false, agg, red, cs, nb, accw, dAlog + kr * C + c,
                              dDs + kr * C + c, ddtb + kr * C + c, ddtw + ((int64_t)kr * C + c) * R);
}

}  // namespace

// x0, x1, dy0, dy1, dx0, dx1: (B,C,L) (orientation 0 row-major pixel order, orientation 1 transposed order); xd0, xd1 as in
// bem_ss2d_scan_strided_f32; dxd0, dxd1: (B,2,R+2,L) contiguous, ZEROED BY THE CALL then accumulated; dAlog, dDs (4C), ddtw (4,C,R),
// ddtb (4,C): accumulated into (the caller's gradient buffers).
extern "C" int bem_ss2d_scan_bwd_f32(const float* x0, const float* x1, const float* xd0, const float* xd1, const float* dy0, const float* dy1,
                                     const float* dtw, const float* dtb, const float* A, const float* Ds, float* dx0, float* dx1, float* dxd0,
                                     float* dxd1, float* dAlog, float* dDs, float* ddtw, float* ddtb, int B, int C, int L, int R,
                                     int64_t xd0_bstride, int64_t xd1_bstride, void* stream) {
    BEM_REQUIRE(x0 && x1 && xd0 && xd1 && dy0 && dy1 && dtw && dtb && A && Ds && dx0 && dx1 && dxd0 && dxd1 && dAlog && dDs && ddtw && ddtb,
                "ss2d_scan_bwd: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && L >= 0 && R >= 1 && R <= 16 && (int64_t)B * C * 2 < (1ll << 31), "ss2d_scan_bwd: bad shape B=%d C=%d L=%d R=%d", B, C, L, R);
    const int64_t xbs0 = xd0_bstride ? xd0_bstride : (int64_t)2 * (R + 2) * L, xbs1 = xd1_bstride ? xd1_bstride : (int64_t)2 * (R + 2) * L;
    BEM_REQUIRE(xbs0 >= (int64_t)2 * (R + 2) * L && xbs1 >= (int64_t)2 * (R + 2) * L, "ss2d_scan_bwd: x_dbl batch strides");
    BEM_REQUIRE((int64_t)L <= (int64_t)SB_MAXCH * 1024 * 4, "ss2d_scan_bwd: L too long");
    if (B == 0 || L == 0) return BEM_OK;
    if (L % 4 == 0)
        BEM_REQUIRE((((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)xd0 | (uintptr_t)xd1 | (uintptr_t)dy0 | (uintptr_t)dy1 | (uintptr_t)dx0 | (uintptr_t)dx1) & 15) == 0 &&
                    xbs0 % 4 == 0 && xbs1 % 4 == 0, "ss2d_scan_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const size_t nd = sizeof(float) * (size_t)B * 2 * (R + 2) * L;
    if (hipMemsetAsync(dxd0, 0, nd, s) != hipSuccess || hipMemsetAsync(dxd1, 0, nd, s) != hipSuccess) return bem_check_launch("ss2d_scan_bwd memset");
    const int grid = C * B * 2;
    if (L <= 1024)
        ss2d_scan_bwd_kernel<256, 4><<<grid, 256, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    else
        ss2d_scan_bwd_kernel<1024, 4><<<grid, 1024, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    return bem_check_launch("ss2d_scan_bwd");
}
