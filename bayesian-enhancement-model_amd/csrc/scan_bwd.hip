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
#include <cstdlib>

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


// Planes of at most 256 pixels (Stage I: 8x8 .. 2x2 under 128x128 crops): one wavefront holds a whole row, and what costs is the atomic
// traffic of every channel adding its (R + 2) x L x_dbl gradients onto the same addresses.  A workgroup of one wavefront takes CBS channels
// of an (orientation, image) one after the other, collects their x_dbl gradients in LDS (both directions: 2 (R + 2) L floats) and adds
// the sums to memory once -- CBS times fewer global atomics.
template <int CBS>
__global__ __launch_bounds__(64) void ss2d_scan_bwd_small_kernel(
    const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ xd0, const float* __restrict__ xd1,
    const float* __restrict__ dy0, const float* __restrict__ dy1, const float* __restrict__ dtw, const float* __restrict__ dtb,
    const float* __restrict__ A, const float* __restrict__ Ds, float* __restrict__ dx0, float* __restrict__ dx1,
    float* __restrict__ dxd0, float* __restrict__ dxd1, float* __restrict__ dAlog, float* __restrict__ dDs, float* __restrict__ ddtw,
    float* __restrict__ ddtb, int Bn, int C, int L, int R, int64_t xbs0, int64_t xbs1) {
    __shared__ float agg[2];
    __shared__ float red[1];
    __shared__ float cs[SB_MAXCH];
    __shared__ float nb[3];
    __shared__ float accw[16];
    __shared__ float dacc[2 * 18 * 256];
    const int G = (C + CBS - 1) / CBS;
    const int wi = blockIdx.x;
    const int g = wi % G, b = (wi / G) % Bn, o = wi / (G * Bn);
    const int nacc = 2 * (R + 2) * L;
    for (int i = threadIdx.x; i < nacc; i += 64) dacc[i] = 0.f;
    __syncthreads();
    const float* xd = o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0;
    const int kf = o, kr = o + 2;
    for (int ch = 0; ch < CBS; ++ch) {
        const int c = g * CBS + ch;
        if (c >= C) break;                                                   // uniform
        const int64_t row = ((int64_t)b * C + c) * L;
        const float* xr = (o ? x1 : x0) + row;
        const float* dyr = (o ? dy1 : dy0) + row;
        float* dxr = (o ? dx1 : dx0) + row;
        ss2d_dir_bwd<64, 4, false>(xr, dyr, xd, dacc, dxr, dtw + ((int64_t)kf * C + c) * R, dtb[kf * C + c], A[kf * C + c], Ds[kf * C + c], L, R,
                                   true, agg, red, cs, nb, accw, dAlog + kf * C + c, dDs + kf * C + c, ddtb + kf * C + c,
                                   ddtw + ((int64_t)kf * C + c) * R);
        ss2d_dir_bwd<64, 4, true>(xr, dyr, xd + (int64_t)(R + 2) * L, dacc + (R + 2) * L, dxr, dtw + ((int64_t)kr * C + c) * R,
                                  dtb[kr * C + c], A[kr * C + c], Ds[kr * C + c], L, R, false, agg, red, cs, nb, accw, dAlog + kr * C + c,
                                  dDs + kr * C + c, ddtb + kr * C + c, ddtw + ((int64_t)kr * C + c) * R);
    }
    __syncthreads();
    float* dxd = (o ? dxd1 : dxd0) + (int64_t)b * 2 * (R + 2) * L;
    for (int i = threadIdx.x; i < nacc; i += 64) atomicAdd(dxd + i, dacc[i]);
}

// ------------------------------------------------------------------------------------------------------------------------
// Whole-row, channel-blocked form for L == NT * 4 * T (the planes of 256x256 / 128x128 inputs): the structure of the forward's
// ss2d_scan_rows_kernel.  A workgroup owns CB channels of one (orientation, image); tiles are the outer loop, channels the
// inner one, so the float4 of every x_dbl plane is loaded once per tile and the x_dbl gradient of a tile is summed over the CB
// channels in registers before it is added to memory (CB times fewer atomics, and each atomic wave-instruction covers 256
// contiguous bytes after a wave-private LDS transpose: the full-rate shape).
//   pass 1 (tiles in scan order)   state entering every thread's 4 elements, kept in registers (CB x T floats)
//   pass 2 (tiles in reverse)      replay h from it, then the adjoint recurrence in the form g_t = a_t (C_t dy_t + g_{t+1}),
//                                  dh_t = C_t dy_t + g_{t+1}: its per-thread map needs only the thread's own a's, so the DPP
//                                  wavefront scan + one LDS barrier per (tile, channel) is all the communication there is.
// ------------------------------------------------------------------------------------------------------------------------
template <int NT, int T, int CB, int R>
__global__ __launch_bounds__(NT) void ss2d_scan_bwd_rows_kernel(
    const float* x0, const float* x1, const float* xd0, const float* xd1, const float* dy0, const float* dy1,
    const float* __restrict__ dtw, const float* __restrict__ dtb, const float* __restrict__ A, const float* __restrict__ Ds,
    float* dx0, float* dx1, float* dxd0, float* dxd1, float* __restrict__ dAlog, float* __restrict__ dDs, float* __restrict__ ddtw,
    float* __restrict__ ddtb, int Bn, int C, int64_t xbs0, int64_t xbs1) {
    constexpr int NW = NT / BEM_WAVE, L = NT * 4 * T, NP = 3 + R;
    __shared__ float agg[2][2 * NW];
    __shared__ float tbuf[NT * 4];
    __shared__ float redp[NW][CB * NP];
    extern __shared__ float enter_sm[];                 // [CB][T][NT]: state entering every thread's 4 elements of every tile
    const int G = (C + CB - 1) / CB;
    const int total = gridDim.x, lin = blockIdx.x;
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int g = wi % G, b = (wi / G) % Bn, o = wi / (G * Bn);
    const float* xb = (o ? x1 : x0) + (int64_t)b * C * L;
    const float* dyb = (o ? dy1 : dy0) + (int64_t)b * C * L;
    float* dxb = (o ? dx1 : dx0) + (int64_t)b * C * L;
    const float* xdb = o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0;
    float* dxdb = (o ? dxd1 : dxd0) + (int64_t)b * 2 * (R + 2) * L;
    const int lane = threadIdx.x & (BEM_WAVE - 1), wave = threadIdx.x / BEM_WAVE;
    int slot = 0;
    constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        const float* xd = xdb + (int64_t)dir * (R + 2) * L;
        float* dxd = dxdb + (int64_t)dir * (R + 2) * L;
        const int kd = o + 2 * dir;
        float carry[CB];
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) carry[ch] = 0.f;
        // ---------------- pass 1 ----------------
#pragma unroll 1
        for (int kk = 0; kk < T; ++kk) {
            const int k = dir ? T - 1 - kk : kk;
            const int pos = (k * NT + threadIdx.x) * 4;
            __builtin_amdgcn_sched_barrier(0);
            float4 dq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dq[r] = *reinterpret_cast<const float4*>(xd + (int64_t)r * L + pos);
            const float4 Bq = *reinterpret_cast<const float4*>(xd + (int64_t)R * L + pos);
            // loaded values that meet packed-f32 arithmetic go through a VALU copy first (scripts/isa_audit.py check 2)
            const float Bv[4] = {valu_copy(Bq.x), valu_copy(Bq.y), valu_copy(Bq.z), valu_copy(Bq.w)};
#pragma unroll
            for (int ch = 0; ch < CB; ++ch) {
                const int c = min(g * CB + ch, C - 1);
                __builtin_amdgcn_sched_barrier(0);
                const float4 xq = *reinterpret_cast<const float4*>(xb + (int64_t)c * L + pos);
                const float xv[4] = {valu_copy(xq.x), valu_copy(xq.y), valu_copy(xq.z), valu_copy(xq.w)};
                const float* wd = dtw + ((int64_t)kd * C + c) * R;
                const float bias = dtb[kd * C + c], Ak = A[kd * C + c];
                float z[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = bias;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float w = wd[r];
                    z[0] = fmaf(w, dq[r].x, z[0]); z[1] = fmaf(w, dq[r].y, z[1]);
                    z[2] = fmaf(w, dq[r].z, z[2]); z[3] = fmaf(w, dq[r].w, z[3]);
                }
                float P = 1.f, S = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? 3 - i : i;
                    const float lg = z[e] <= 20.f ? __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z[e] * LOG2E)) : z[e] * LOG2E;
                    const float a = __builtin_amdgcn_exp2f(lg * Ak);
                    S = fmaf(a, S, lg * LN2 * Bv[e] * xv[e]);
                    P = P * a;
                }
                float Pe, Se;
                if (dir) wave_scan_affine<true>(P, S, Pe, Se);
                else wave_scan_affine<false>(P, S, Pe, Se);
                const float hw = dir ? cross_wave_affine<NW, true>(P, S, agg[slot], carry[ch]) : cross_wave_affine<NW, false>(P, S, agg[slot], carry[ch]);
                slot ^= 1;
                enter_sm[(ch * T + k) * NT + threadIdx.x] = fmaf(Pe, hw, Se);
            }
        }
        // ---------------- pass 2 ----------------
        float rq[CB], pacc[CB][NP];
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) {
            rq[ch] = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) pacc[ch][q] = 0.f;
        }
#pragma unroll 1
        for (int kk = T - 1; kk >= 0; --kk) {
            const int k = dir ? T - 1 - kk : kk;
            const int pos = (k * NT + threadIdx.x) * 4;
            __builtin_amdgcn_sched_barrier(0);
            float4 dq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dq[r] = *reinterpret_cast<const float4*>(xd + (int64_t)r * L + pos);
            const float4 Bq = *reinterpret_cast<const float4*>(xd + (int64_t)R * L + pos);
            const float4 Cq = *reinterpret_cast<const float4*>(xd + (int64_t)(R + 1) * L + pos);
            const float Bv[4] = {valu_copy(Bq.x), valu_copy(Bq.y), valu_copy(Bq.z), valu_copy(Bq.w)},
                        Cv[4] = {valu_copy(Cq.x), valu_copy(Cq.y), valu_copy(Cq.z), valu_copy(Cq.w)};
            float acc[R + 2][4];
#pragma unroll
            for (int q = 0; q < R + 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
#pragma unroll
            for (int ch = 0; ch < CB; ++ch) {
                const int c = min(g * CB + ch, C - 1);
                const bool live = g * CB + ch < C;              // a partial last group recomputes channel C - 1 without contributing
                __builtin_amdgcn_sched_barrier(0);
                const float4 xq = *reinterpret_cast<const float4*>(xb + (int64_t)c * L + pos);
                const float4 gq = *reinterpret_cast<const float4*>(dyb + (int64_t)c * L + pos);
                const float xv[4] = {valu_copy(xq.x), valu_copy(xq.y), valu_copy(xq.z), valu_copy(xq.w)},
                            dy[4] = {valu_copy(gq.x), valu_copy(gq.y), valu_copy(gq.z), valu_copy(gq.w)};
                const float* wd = dtw + ((int64_t)kd * C + c) * R;
                const float bias = dtb[kd * C + c], Ak = A[kd * C + c], Dk = Ds[kd * C + c];
                float wr[R];
#pragma unroll
                for (int r = 0; r < R; ++r) wr[r] = wd[r];
                float z[4], lg[4], a[4], bb[4], h[4], br[4], dh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = bias;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    z[0] = fmaf(wr[r], dq[r].x, z[0]); z[1] = fmaf(wr[r], dq[r].y, z[1]);
                    z[2] = fmaf(wr[r], dq[r].z, z[2]); z[3] = fmaf(wr[r], dq[r].w, z[3]);
                }
                float hh = enter_sm[(ch * T + k) * NT + threadIdx.x];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? 3 - i : i;
                    lg[e] = z[e] <= 20.f ? __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z[e] * LOG2E)) : z[e] * LOG2E;
                    a[e] = __builtin_amdgcn_exp2f(lg[e] * Ak);
                    bb[e] = lg[e] * LN2 * Bv[e] * xv[e];
                    hh = fmaf(a[e], hh, bb[e]);
                    h[e] = hh;
                    br[e] = Cv[e] * dy[e];
                }
                // adjoint: visit the elements in reverse scan order, g' = a_e (br_e + g)
                float P = 1.f, S = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? i : 3 - i;
                    S = a[e] * (S + br[e]);
                    P = P * a[e];
                }
                float Pe, Se;
                if (dir) wave_scan_affine<false>(P, S, Pe, Se);
                else wave_scan_affine<true>(P, S, Pe, Se);
                const float qw = dir ? cross_wave_affine<NW, false>(P, S, agg[slot], rq[ch]) : cross_wave_affine<NW, true>(P, S, agg[slot], rq[ch]);
                slot ^= 1;
                float gg = fmaf(Pe, qw, Se);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? i : 3 - i;
                    dh[e] = br[e] + gg;
                    gg = a[e] * dh[e];
                }
                const float lv = live ? 1.f : 0.f;
                float dxv[4], dz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dl = lg[e] * LN2;
                    const float hm = h[e] - bb[e];
                    const float dhd = dh[e] * dl * lv;
                    dxv[e] = fmaf(dhd, Bv[e], Dk * dy[e]);
                    const float ddl = dh[e] * fmaf(Bv[e], xv[e], Ak * hm);
                    const float sg = z[e] <= 20.f ? 1.f - __builtin_amdgcn_exp2f(-lg[e]) : 1.f;      // sigmoid(z) = 1 - 2^-log2(1 + e^z)
                    dz[e] = ddl * sg * lv;
                    acc[R][e] = fmaf(dhd, xv[e], acc[R][e]);
                    acc[R + 1][e] = fmaf(dy[e] * lv, h[e], acc[R + 1][e]);
                    pacc[ch][0] = fmaf(dhd, hm, pacc[ch][0]);
                    pacc[ch][1] = fmaf(dy[e], xv[e], pacc[ch][1]);
                    pacc[ch][2] += dz[e];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r][0] = fmaf(dz[0], wr[r], acc[r][0]); acc[r][1] = fmaf(dz[1], wr[r], acc[r][1]);
                    acc[r][2] = fmaf(dz[2], wr[r], acc[r][2]); acc[r][3] = fmaf(dz[3], wr[r], acc[r][3]);
                    pacc[ch][3 + r] += fmaf(dz[0], dq[r].x, fmaf(dz[1], dq[r].y, fmaf(dz[2], dq[r].z, dz[3] * dq[r].w)));
                }
                if (live) {
                    float* dp = dxb + (int64_t)c * L + pos;
                    if (dir) {
                        const float4 pv = *reinterpret_cast<const float4*>(dp);
                        dxv[0] += pv.x; dxv[1] += pv.y; dxv[2] += pv.z; dxv[3] += pv.w;
                    }
                    *reinterpret_cast<float4*>(dp) = make_float4(dxv[0], dxv[1], dxv[2], dxv[3]);
                }
            }
            // tile flush: wave-private LDS transpose, then four 256-byte-contiguous atomic wave-instructions per plane
            float* tw = tbuf + wave * 256;
            float* dbase = dxd + (int64_t)(k * NT + wave * BEM_WAVE) * 4;
#pragma unroll
            for (int q = 0; q < R + 2; ++q) {
                *reinterpret_cast<float4*>(tw + lane * 4) = make_float4(acc[q][0], acc[q][1], acc[q][2], acc[q][3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dbase + (int64_t)q * L + 64 * j + lane, tw[64 * j + lane]);
            }
        }
        // parameter gradients of this direction: wave sums -> LDS -> one atomic per value
#pragma unroll
        for (int ch = 0; ch < CB; ++ch)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                float v = pacc[ch][q];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, BEM_WAVE);
                if (lane == 0) redp[wave][ch * NP + q] = v;
            }
        __syncthreads();
        if (threadIdx.x < CB * NP) {
            const int ch = threadIdx.x / NP, q = threadIdx.x % NP, c = g * CB + ch;
            if (c < C) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += redp[w][threadIdx.x];
                const int kc = kd * C + c;
                if (q == 0) atomicAdd(dAlog + kc, v * A[kc]);
                else if (q == 1) atomicAdd(dDs + kc, v);
                else if (q == 2) atomicAdd(ddtb + kc, v);
                else atomicAdd(ddtw + (int64_t)kc * R + (q - 3), v);
            }
        }
        __syncthreads();
    }
}

template <int NT, int T, int CB, int R>
static int launch_bwd_rows(const float* x0, const float* x1, const float* xd0, const float* xd1, const float* dy0, const float* dy1, const float* dtw,
                           const float* dtb, const float* A, const float* Ds, float* dx0, float* dx1, float* dxd0, float* dxd1, float* dAlog, float* dDs,
                           float* ddtw, float* ddtb, int B, int C, int64_t xbs0, int64_t xbs1, hipStream_t s) {
    const int G = (C + CB - 1) / CB;
    constexpr size_t lds = sizeof(float) * CB * T * NT;
    static_assert(lds + sizeof(float) * (NT * 4 + 64 + (NT / 64) * CB * (3 + R)) <= 160 * 1024, "LDS budget");
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ss2d_scan_bwd_rows_kernel<NT, T, CB, R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    ss2d_scan_bwd_rows_kernel<NT, T, CB, R><<<G * B * 2, NT, lds, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs,
                                                                      ddtw, ddtb, B, C, xbs0, xbs1);
    return bem_check_launch("ss2d_scan_bwd(rows)");
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
    static const bool fast = !(getenv("BEM_SCAN_BWD_ROWS") && atoi(getenv("BEM_SCAN_BWD_ROWS")) == 0);
    if (fast) {
        // whole-row channel-blocked forms for the plane sizes / dt_ranks of the shipped configuration (n_feat 40: R = 3 / 5 / 10)
#define BEM_BWD_ROWS(NT, T, CB, RR) return launch_bwd_rows<NT, T, CB, RR>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, xbs0, xbs1, s)
        if (L == 16384 && R == 3) BEM_BWD_ROWS(512, 8, 4, 3);
        if (L == 4096 && R == 3) BEM_BWD_ROWS(256, 4, 4, 3);
        if (L == 4096 && R == 5) BEM_BWD_ROWS(256, 4, 4, 5);
        if (L == 1024 && R == 5) BEM_BWD_ROWS(256, 1, 4, 5);
        if (L == 1024 && R == 10) BEM_BWD_ROWS(256, 1, 2, 10);
        if (L == 256 && R == 10) BEM_BWD_ROWS(64, 1, 2, 10);
        if (L == 1024 && R == 1) BEM_BWD_ROWS(256, 1, 4, 1);
        if (L == 1024 && R == 2) BEM_BWD_ROWS(256, 1, 2, 2);
#undef BEM_BWD_ROWS
    }
    const int grid = C * B * 2;
    if (L <= 32)          // 4x4 and 2x2 planes: a workgroup per channel (the atomics are few, the channel loop below would only serialise)
        ss2d_scan_bwd_kernel<64, 4><<<grid, 64, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    else if (L <= 256) {  // 8x8 .. 16x16 planes: one wavefront per row, 4 channels per workgroup (181 -> 114 us at L = 64, C = 160, B = 8)
        constexpr int CBS = 4;
        ss2d_scan_bwd_small_kernel<CBS><<<cdiv(C, CBS) * B * 2, 64, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    }
    else if (L <= 1024)
        ss2d_scan_bwd_kernel<256, 4><<<grid, 256, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    else
        ss2d_scan_bwd_kernel<1024, 4><<<grid, 1024, 0, s>>>(x0, x1, xd0, xd1, dy0, dy1, dtw, dtb, A, Ds, dx0, dx1, dxd0, dxd1, dAlog, dDs, ddtw, ddtb, B, C, L, R, xbs0, xbs1);
    return bem_check_launch("ss2d_scan_bwd");
}
