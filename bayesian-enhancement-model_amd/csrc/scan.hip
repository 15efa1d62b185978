// Selective scan kernels for CDNA4 (wave64).
//   - bem_selective_scan_fwd_f32 : operator-seam replacement of selective_scan_cuda_oflex.fwd
//   - bem_ss2d_scan_f32          : fused x_proj-rows -> dt_proj -> softplus -> 4-direction scan (N = 1)
//   - bem_cross_scan / merge     : operator-seam replacements of the Triton cross scan / merge
//
// Scan structure: every thread owns E consecutive sequence positions, folds them into an affine map
// h -> P*h + S, the 64 lanes of a wave compose their maps with wavefront shuffles (Hillis-Steele,
// 6 steps), waves exchange their aggregates through LDS and a running carry links successive chunks.
// The reverse directions use the mirrored lane / wave / element order of the same code.
#include "scan_common.h"
#include <cstdlib>

namespace {


// ------------------------------------------------------------------------------------------------
// General selective scan (any dstate, groups): one workgroup per (batch, channel) row.
// ------------------------------------------------------------------------------------------------
template <int NT, int E, typename TI>
__global__ __launch_bounds__(NT) void selective_scan_fwd_kernel(
    const TI* __restrict__ u, const TI* __restrict__ delta, const float* __restrict__ A,
    const TI* __restrict__ Bm, const TI* __restrict__ Cm, const float* __restrict__ D,
    const float* __restrict__ dbias, float* __restrict__ out, int dim, int L, int dstate, int ngroups,
    int softplus) {
    __shared__ float agg[2 * (NT / BEM_WAVE)];
    __shared__ float carry_s[256];
    const int d = blockIdx.x, b = blockIdx.y;
    const int g = d / (dim / ngroups);
    const int64_t row = ((int64_t)b * dim + d) * L;
    const TI* ur = u + row;
    const TI* dr = delta + row;
    float* yr = out + row;
    const float bias = dbias ? dbias[d] : 0.f;
    const float Dd = D ? D[d] : 0.f;
    const bool vec = (L % 4 == 0);
    for (int n = threadIdx.x; n < dstate; n += NT) carry_s[n] = 0.f;
    __syncthreads();
    constexpr int CH = NT * E;
    for (int64_t c0 = 0; c0 < L; c0 += CH) {
        const int64_t t0 = c0 + (int64_t)threadIdx.x * E;
        float uu[E], dt[E], y[E];
        load_row<E>(ur, t0, L, vec, uu);
        load_row<E>(dr, t0, L, vec, dt);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dt[e] += bias;
            if (softplus) dt[e] = bem_softplus(dt[e]);
            y[e] = Dd * uu[e];
        }
        for (int n = 0; n < dstate; ++n) {
            const float An = A[(int64_t)d * dstate + n];
            const int64_t bc = (((int64_t)b * ngroups + g) * dstate + n) * L;
            float Bv[E], Cv[E], a[E], bb[E], h[E];
            load_row<E>(Bm + bc, t0, L, vec, Bv);
            load_row<E>(Cm + bc, t0, L, vec, Cv);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ok = t0 + e < L;
                a[e] = ok ? bem_fexp(dt[e] * An) : 1.f;
                bb[e] = ok ? dt[e] * Bv[e] * uu[e] : 0.f;
            }
            float carry = carry_s[n];
            block_scan_affine<NT, E, false>(a, bb, h, carry, agg);
            if (threadIdx.x == 0) carry_s[n] = carry;
#pragma unroll
            for (int e = 0; e < E; ++e) y[e] += Cv[e] * h[e];
        }
        store_row<E>(yr, t0, L, vec, y);
        __syncthreads();   // carry_s visible to all threads before the next chunk
    }
}

// ------------------------------------------------------------------------------------------------
// General selective scan, backward (replaces selective_scan_cuda_oflex.bwd, selective_scan_bwd_kernel_oflex.cuh:73-289).
// One workgroup per (batch, channel) row.
//   pass 1 (chunks ascending)  recompute the forward states, keep the state entering every chunk in `ws`
//   pass 2 (chunks descending) per chunk and state n: rebuild h_t from the saved carry (forward block scan), then
//           dh_t = C_t dy_t + a_{t+1} dh_{t+1} by the mirrored (reverse) block scan, then
//             du_t   = D dy_t + sum_n dh dt B            ddt_t = sum_n dh (B u + A (h - dt B u))
//             dB_t,n += dh dt u  , dC_t,n += dy h        (atomics: shared by the channels of a group, like the reference)
//             dA_n  += dh dt (h - dt B u)                dD += dy u
//           ddelta_t = ddt_t * sigmoid(delta + bias) (softplus), ddelta_bias += ddelta_t
// a_{t+1} of a thread's last element is recomputed from delta[t+1] (one extra exp per thread) instead of being
// exchanged between lanes.
// ------------------------------------------------------------------------------------------------

template <int NT, int E>
__global__ __launch_bounds__(NT) void selective_scan_bwd_kernel(
    const float* __restrict__ u, const float* __restrict__ delta, const float* __restrict__ A,
    const float* __restrict__ Bm, const float* __restrict__ Cm, const float* __restrict__ D,
    const float* __restrict__ dbias, const float* __restrict__ dout, float* __restrict__ ws,
    float* __restrict__ du, float* __restrict__ ddelta, float* __restrict__ dA, float* __restrict__ dB,
    float* __restrict__ dC, float* __restrict__ dD, float* __restrict__ ddbias, int dim, int L, int dstate,
    int ngroups, int softplus) {
    __shared__ float agg[2 * (NT / BEM_WAVE)];
    __shared__ float red[NT / BEM_WAVE];
    __shared__ float carry_s[256];
    const int d = blockIdx.x, b = blockIdx.y;
    const int g = d / (dim / ngroups);
    const int64_t row = ((int64_t)b * dim + d) * L;
    const float* ur = u + row;
    const float* dr = delta + row;
    const float* gr = dout + row;
    const float bias = dbias ? dbias[d] : 0.f;
    const float Dd = D ? D[d] : 0.f;
    const bool vec = (L % 4 == 0);
    constexpr int CH = NT * E;
    const int nchunks = (L + CH - 1) / CH;
    float* wsr = ws + ((int64_t)b * dim + d) * nchunks * dstate;

    auto dt_of = [&](float dl) { dl += bias; return softplus ? bem_softplus(dl) : dl; };

    // ---- pass 1: state entering every chunk ----
    for (int n = threadIdx.x; n < dstate; n += NT) carry_s[n] = 0.f;
    __syncthreads();
    for (int j = 0; j < nchunks; ++j) {
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        float uu[E], dt[E];
        load_row<E>(ur, t0, L, vec, uu);
        load_row<E>(dr, t0, L, vec, dt);
#pragma unroll
        for (int e = 0; e < E; ++e) dt[e] = dt_of(dt[e]);
        for (int n = 0; n < dstate; ++n) {
            const float An = A[(int64_t)d * dstate + n];
            float Bv[E], a[E], bb[E], h[E];
            load_row<E>(Bm + (((int64_t)b * ngroups + g) * dstate + n) * L, t0, L, vec, Bv);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ok = t0 + e < L;
                a[e] = ok ? bem_fexp(dt[e] * An) : 1.f;
                bb[e] = ok ? dt[e] * Bv[e] * uu[e] : 0.f;
            }
            float carry = carry_s[n];
            if (threadIdx.x == 0) wsr[(int64_t)j * dstate + n] = carry;
            block_scan_affine<NT, E, false>(a, bb, h, carry, agg);
            if (threadIdx.x == 0) carry_s[n] = carry;
        }
        __syncthreads();
    }
    // ---- pass 2 ----
    for (int n = threadIdx.x; n < dstate; n += NT) carry_s[n] = 0.f;     // dh entering from the right
    __syncthreads();                                                      // also orders the ws stores of thread 0 before its re-reads
    float accD = 0.f, accBias = 0.f;
    for (int j = nchunks - 1; j >= 0; --j) {
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        float uu[E], dl[E], dt[E], dy[E], duv[E], ddt[E];
        load_row<E>(ur, t0, L, vec, uu);
        load_row<E>(dr, t0, L, vec, dl);
        load_row<E>(gr, t0, L, vec, dy);
        const float dl_next = (t0 + E < L) ? dr[t0 + E] : 0.f;
        const float dt_next = dt_of(dl_next);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            dt[e] = dt_of(dl[e]);
            duv[e] = Dd * dy[e];
            ddt[e] = 0.f;
            accD += (t0 + e < L) ? dy[e] * uu[e] : 0.f;
        }
        for (int n = 0; n < dstate; ++n) {
            const float An = A[(int64_t)d * dstate + n];
            const int64_t bc = (((int64_t)b * ngroups + g) * dstate + n) * L;
            float Bv[E], Cv[E], a[E], bb[E], h[E], ar[E], br[E], dh[E];
            load_row<E>(Bm + bc, t0, L, vec, Bv);
            load_row<E>(Cm + bc, t0, L, vec, Cv);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ok = t0 + e < L;
                a[e] = ok ? bem_fexp(dt[e] * An) : 1.f;
                bb[e] = ok ? dt[e] * Bv[e] * uu[e] : 0.f;
            }
            float cf = wsr[(int64_t)j * dstate + n];
            block_scan_affine<NT, E, false>(a, bb, h, cf, agg);
            // reverse: dh_t = a_{t+1} dh_{t+1} + C_t dy_t
            const float a_next = (t0 + E < L) ? bem_fexp(dt_next * An) : 1.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool ok = t0 + e < L;
                ar[e] = (e + 1 < E) ? a[e + 1] : a_next;
                if (!(t0 + e + 1 < L)) ar[e] = 1.f;
                br[e] = ok ? Cv[e] * dy[e] : 0.f;
            }
            float cr = carry_s[n];
            block_scan_affine<NT, E, true>(ar, br, dh, cr, agg);
            if (threadIdx.x == 0) carry_s[n] = cr;
            float accA = 0.f;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (t0 + e < L) {
                    const float hm = h[e] - bb[e];                     // a_t * h_{t-1}
                    duv[e] = fmaf(dh[e] * dt[e], Bv[e], duv[e]);
                    ddt[e] += dh[e] * (Bv[e] * uu[e] + An * hm);
                    accA = fmaf(dh[e] * dt[e], hm, accA);
                    atomicAdd(dB + bc + t0 + e, dh[e] * dt[e] * uu[e]);
                    atomicAdd(dC + bc + t0 + e, dy[e] * h[e]);
                }
            }
            accA = block_reduce_sum<NT>(accA, red);
            if (threadIdx.x == 0) atomicAdd(dA + (int64_t)d * dstate + n, accA);
        }
        float dd[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float z = dl[e] + bias;
            const float sg = (softplus && z <= 20.f) ? 1.f / (1.f + bem_fexp(-z)) : 1.f;
            dd[e] = ddt[e] * sg;
            accBias += (t0 + e < L) ? dd[e] : 0.f;
        }
        store_row<E>(du + row, t0, L, vec, duv);
        store_row<E>(ddelta + row, t0, L, vec, dd);
        __syncthreads();
    }
    accD = block_reduce_sum<NT>(accD, red);
    accBias = block_reduce_sum<NT>(accBias, red);
    if (threadIdx.x == 0) {
        if (dD) atomicAdd(dD + d, accD);
        if (ddbias) atomicAdd(ddbias + d, accBias);
    }
}

// ------------------------------------------------------------------------------------------------
// Fused SS2D scan (N = 1): grid (C, B, 2 orientations).  Each workgroup owns one (b, c) sequence of
// one orientation and runs its forward direction (k = o) and reverse direction (k = o + 2).
// ------------------------------------------------------------------------------------------------
template <int E>
__device__ __forceinline__ void ss2d_coeffs(const float* __restrict__ xd /* (R+2, L) of one direction */,
                                            const float* __restrict__ wdt /* (R) */, float dtb, float Ak,
                                            const float (&x)[E], int64_t t0, int L, int R, bool vec,
                                            float (&a)[E], float (&b)[E], float (&cv)[E]) {
    float dt[E];
#pragma unroll
    for (int e = 0; e < E; ++e) dt[e] = 0.f;
    for (int r = 0; r < R; ++r) {
        float v[E];
        load_row<E>(xd + (int64_t)r * L, t0, L, vec, v);
        const float w = wdt[r];
#pragma unroll
        for (int e = 0; e < E; ++e) dt[e] = fmaf(w, v[e], dt[e]);
    }
    float Bv[E];
    load_row<E>(xd + (int64_t)R * L, t0, L, vec, Bv);
    load_row<E>(xd + (int64_t)(R + 1) * L, t0, L, vec, cv);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool ok = t0 + e < L;
        const float dl = bem_softplus(dt[e] + dtb);
        a[e] = ok ? bem_fexp(dl * Ak) : 1.f;
        b[e] = ok ? dl * Bv[e] * x[e] : 0.f;
    }
}

template <int NT, int E>
__global__ __launch_bounds__(NT) void ss2d_scan_kernel(
    const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ xd0,
    const float* __restrict__ xd1, const float* __restrict__ dtw, const float* __restrict__ dtb,
    const float* __restrict__ A, const float* __restrict__ Ds, float* __restrict__ y0,
    float* __restrict__ y1, int Bn, int C, int L, int R, int64_t xbs0, int64_t xbs1) {
    __shared__ float agg[2 * (NT / BEM_WAVE)];
    // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs, so give each XCD a contiguous range of
    // (orientation, image, channel) work items -- the C channel rows of one (o, b) share its 2*(R+2) x_dbl planes in that XCD's L2.
    const int total = gridDim.x, lin = blockIdx.x;
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int c = wi % C, b = (wi / C) % Bn, o = wi / (C * Bn);
    const float* xr = (o ? x1 : x0) + ((int64_t)b * C + c) * L;
    const float* xd = (o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0);
    float* yr = (o ? y1 : y0) + ((int64_t)b * C + c) * L;
    const int kf = o, kr = o + 2;
    const float* wf = dtw + ((int64_t)kf * C + c) * R;
    const float* wr = dtw + ((int64_t)kr * C + c) * R;
    const float bf = dtb[kf * C + c], br = dtb[kr * C + c];
    const float Af = A[kf * C + c], Ar = A[kr * C + c];
    const float Df = Ds[kf * C + c], Dr = Ds[kr * C + c];
    const bool vec = (L % 4 == 0);
    constexpr int CH = NT * E;
    const int nchunks = (L + CH - 1) / CH;
    const bool single = nchunks == 1;
    float yacc[E];
    // forward direction, chunks ascending
    float carry = 0.f;
    for (int j = 0; j < nchunks; ++j) {
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        float x[E], a[E], bb[E], cv[E], h[E];
        load_row<E>(xr, t0, L, vec, x);
        ss2d_coeffs<E>(xd, wf, bf, Af, x, t0, L, R, vec, a, bb, cv);
        block_scan_affine<NT, E, false>(a, bb, h, carry, agg);
#pragma unroll
        for (int e = 0; e < E; ++e) yacc[e] = fmaf(cv[e], h[e], Df * x[e]);
        if (!single) store_row<E>(yr, t0, L, vec, yacc);
    }
    // reverse direction, chunks descending; adds onto the forward result
    carry = 0.f;
    for (int j = nchunks - 1; j >= 0; --j) {
        const int64_t t0 = (int64_t)j * CH + (int64_t)threadIdx.x * E;
        float x[E], a[E], bb[E], cv[E], h[E], yv[E];
        load_row<E>(xr, t0, L, vec, x);
        ss2d_coeffs<E>(xd + (int64_t)(R + 2) * L, wr, br, Ar, x, t0, L, R, vec, a, bb, cv);
        block_scan_affine<NT, E, true>(a, bb, h, carry, agg);
        if (!single) load_row<E>(yr, t0, L, vec, yacc);   // this thread's own earlier stores
#pragma unroll
        for (int e = 0; e < E; ++e) yv[e] = yacc[e] + fmaf(cv[e], h[e], Dr * x[e]);
        store_row<E>(yr, t0, L, vec, yv);
    }
}

// Whole-row form for L == NT * E (the 128x128 / 64x64 / 32x32 planes of a 256x256 image): x, the forward result and the
// reverse result stay in registers, so x is read once and y written once; no bounds masks.  C_t is reloaded from the
// (L2-resident) x_dbl plane after the scan instead of being held across it.
template <int E, int RT>      // RT = dt_rank when known at compile time (its plane loads then issue together), 0 = runtime R
__device__ __forceinline__ void ss2d_coeffs_full(const float* __restrict__ xd, const float* __restrict__ wdt, float dtb,
                                                 float Ak, const float (&x)[E], int t0, int L, int Rr, float (&a)[E],
                                                 float (&b)[E]) {
    const int R = RT ? RT : Rr;
#pragma unroll
    for (int e = 0; e < E; ++e) a[e] = dtb;
#pragma unroll
    for (int r = 0; r < (RT ? RT : R); ++r) {
        const float w = wdt[r];
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(xd + (int64_t)r * L + t0 + i);
            a[i] = fmaf(w, q.x, a[i]); a[i + 1] = fmaf(w, q.y, a[i + 1]);
            a[i + 2] = fmaf(w, q.z, a[i + 2]); a[i + 3] = fmaf(w, q.w, a[i + 3]);
        }
    }
#pragma unroll
    for (int i = 0; i < E; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(xd + (int64_t)R * L + t0 + i);
        const float bv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dl = bem_softplus(a[i + j]);
            a[i + j] = bem_fexp(dl * Ak);
            b[i + j] = dl * bv[j] * x[i + j];
        }
    }
}

template <int NT, int E, int RT>
__global__ __launch_bounds__(NT) void ss2d_scan_full_kernel(
    const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ xd0,
    const float* __restrict__ xd1, const float* __restrict__ dtw, const float* __restrict__ dtb,
    const float* __restrict__ A, const float* __restrict__ Ds, float* __restrict__ y0,
    float* __restrict__ y1, int Bn, int C, int Rr, int64_t xbs0, int64_t xbs1) {
    __shared__ float agg[2 * (NT / BEM_WAVE)];
    constexpr int L = NT * E;
    const int R = RT ? RT : Rr;
    const int total = gridDim.x, lin = blockIdx.x;      // XCD-aware order, as in ss2d_scan_kernel
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int c = wi % C, b = (wi / C) % Bn, o = wi / (C * Bn);
    const float* xr = (o ? x1 : x0) + ((int64_t)b * C + c) * L;
    const float* xd = (o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0);
    float* yr = (o ? y1 : y0) + ((int64_t)b * C + c) * L;
    const int kf = o, kr = o + 2;
    const int t0 = threadIdx.x * E;
    float x[E], y[E];
#pragma unroll
    for (int i = 0; i < E; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(xr + t0 + i);
        x[i] = q.x; x[i + 1] = q.y; x[i + 2] = q.z; x[i + 3] = q.w;
    }
    {
        float a[E], bb[E];
        ss2d_coeffs_full<E, RT>(xd, dtw + ((int64_t)kf * C + c) * R, dtb[kf * C + c], A[kf * C + c], x, t0, L, R, a, bb);
        float carry = 0.f;
        float hh = block_scan_enter<NT, E, false>(a, bb, carry, agg);
        const float Df = Ds[kf * C + c];
        const float* cp = xd + (int64_t)(R + 1) * L + t0;
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(cp + i);
            const float cv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hh = a[i + j] * hh + bb[i + j];
                y[i + j] = fmaf(cv[j], hh, Df * x[i + j]);
            }
        }
    }
    {
        float a[E], bb[E];
        const float* xdr = xd + (int64_t)(R + 2) * L;
        ss2d_coeffs_full<E, RT>(xdr, dtw + ((int64_t)kr * C + c) * R, dtb[kr * C + c], A[kr * C + c], x, t0, L, R, a, bb);
        float carry = 0.f;
        float hh = block_scan_enter<NT, E, true>(a, bb, carry, agg);
        const float Dr = Ds[kr * C + c];
        const float* cp = xdr + (int64_t)(R + 1) * L + t0;
#pragma unroll
        for (int i = E - 4; i >= 0; i -= 4) {
            const float4 q = *reinterpret_cast<const float4*>(cp + i);
            const float cv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                hh = a[i + j] * hh + bb[i + j];
                y[i + j] += fmaf(cv[j], hh, Dr * x[i + j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < E; i += 4)
        *reinterpret_cast<float4*>(yr + t0 + i) = make_float4(y[i], y[i + 1], y[i + 2], y[i + 3]);
}

// ------------------------------------------------------------------------------------------------
// Channel-blocked whole-row form, L == NT * 4 * T.  The per-channel kernels above re-read the 2 (R + 2) x_dbl planes
// of an image from L2 once per channel (C times), which makes them L2-bandwidth bound; here a workgroup owns CB
// channels of one (orientation, image) and every thread keeps the float4 of each x_dbl plane it needs in registers
// while it walks the CB channels.  Lane l of a tile owns 4 consecutive positions, so a wavefront load is one
// contiguous 1 KB segment; the wavefront scan runs on DPP row shifts / row broadcasts (no LDS traffic); y of both
// directions accumulates in registers (x read twice, y written once).
// ------------------------------------------------------------------------------------------------

// TR: orientation 1 works on the ROW-MAJOR planes too (x1 = x0's tensor, y1 written row-major): a workgroup of that
// orientation stages its CB planes in LDS (row pitch W + 1), reads its column-major scan positions from there and sends
// its result back through the same buffer, so neither the transposed copy of x nor the transposed y exist in HBM.
template <int NT, int T, int CB, int R, int MINW, bool TR, int ORI = -1>
__global__ __launch_bounds__(NT, MINW) void ss2d_scan_rows_kernel(
    // x / x_dbl deliberately not __restrict__: their loads must stay behind the barrier of the channel step they belong
    // to (as invariant loads the compiler hoists all CB * T * 2 steps' loads to the top and runs out of registers)
    const float* x0, const float* x1, const float* xd0, const float* xd1, const float* __restrict__ dtw,
    const float* __restrict__ dtb, const float* __restrict__ A, const float* __restrict__ Ds, float* y0, float* y1,
    int Bn, int C, int64_t xbs0, int64_t xbs1, int Himg) {
    constexpr int NW = NT / BEM_WAVE, L = NT * 4 * T;
    constexpr int NSLOT = CB > 1 ? CB : 2;
    __shared__ float agg[NSLOT][2 * NW];
    extern __shared__ float plane_sm[];                  // TR: CB planes, Himg rows of (Wimg + 1) floats
    const int G = (C + CB - 1) / CB;
    const int total = gridDim.x, lin = blockIdx.x;      // XCD-aware order: the channel groups of one (o, b) share an L2
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int g = wi % G, b = (wi / G) % Bn, o = ORI < 0 ? wi / (G * Bn) : ORI;
    const float* xb = (o ? x1 : x0) + (int64_t)b * C * L;
    const float* xdb = (o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0);
    float* yb = (o ? y1 : y0) + (int64_t)b * C * L;
    const int lane = threadIdx.x & (BEM_WAVE - 1), wave = threadIdx.x / BEM_WAVE;
    const bool via_lds = TR && o == 1;                   // uniform per workgroup (a compile-time constant when ORI >= 0)
    // LDS plane layout: element (row, col) at (row >> 2) * (4 pitch + 1) + (row & 3) * pitch + col, pitch = W + 1.  A lane reads
    // 4 consecutive rows of one column (rows 4n .. 4n + 3): the extra +1 per group of 4 rows makes the lane stride
    // 4 pitch + 1 (odd), so the 32 lanes of a half-wave hit 32 different banks.
    const int Wimg = TR ? L / Himg : 0, pitch = Wimg + 1, gpitch = 4 * pitch + 1, psz = (Himg >> 2) * gpitch;
    if (via_lds) {
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) {
            const float* src = xb + (int64_t)min(g * CB + ch, C - 1) * L;
            float* pl = plane_sm + ch * psz;
            for (int i = threadIdx.x * 4; i < L; i += NT * 4) {          // W % 4 == 0: the 4 pixels share a row
                const float4 v = *reinterpret_cast<const float4*>(src + i);
                const int row = i / Wimg, col = i - row * Wimg;
                float* d = pl + (row >> 2) * gpitch + (row & 3) * pitch + col;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        __syncthreads();
    }
    float4 y[CB][T];
    int slot = 0;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
        const float* xd = xdb + (int64_t)dir * (R + 2) * L;
        const int kd = o + 2 * dir;
        float carry[CB];
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) carry[ch] = 0.f;
#pragma unroll
        for (int kk = 0; kk < T; ++kk) {
            const int k = dir ? T - 1 - kk : kk;
            const int pos = (k * NT + threadIdx.x) * 4;
            __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from pulling every later step's loads up here
            float4 dq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) dq[r] = *reinterpret_cast<const float4*>(xd + (int64_t)r * L + pos);
            const float4 Bq = *reinterpret_cast<const float4*>(xd + (int64_t)R * L + pos);
            // orientation-1-only launch: C_t is loaded after the scan of a channel instead of being held across it (4 of the 64
            // registers that form has; the plane is L2-resident)
            constexpr bool LATE_C = TR && ORI == 1;
            float4 Cq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!LATE_C) Cq = *reinterpret_cast<const float4*>(xd + (int64_t)(R + 1) * L + pos);
            const float Bv[4] = {Bq.x, Bq.y, Bq.z, Bq.w};
            float Cv[4] = {Cq.x, Cq.y, Cq.z, Cq.w};
            // column-major scan position pos = col * H + row (H % 4 == 0: the 4 positions share a column)
            const int pcol = via_lds ? pos / Himg : 0, prow = via_lds ? pos - pcol * Himg : 0;
            const int lofs = (prow >> 2) * gpitch + pcol;          // prow % 4 == 0
#pragma unroll
            for (int ch = 0; ch < CB; ++ch) {
                const int c = min(g * CB + ch, C - 1);          // a partial last group recomputes channel C - 1 (not stored)
                __builtin_amdgcn_sched_barrier(0);
                float xv[4];
                if (via_lds) {
                    const float* pl = plane_sm + ch * psz + lofs;
                    xv[0] = pl[0]; xv[1] = pl[pitch]; xv[2] = pl[2 * pitch]; xv[3] = pl[3 * pitch];
                } else {
                    const float4 xq = *reinterpret_cast<const float4*>(xb + (int64_t)c * L + pos);
                    xv[0] = xq.x; xv[1] = xq.y; xv[2] = xq.z; xv[3] = xq.w;
                }
                const float* wd = dtw + ((int64_t)kd * C + c) * R;
                const float bias = dtb[kd * C + c], Ak = A[kd * C + c], Dk = Ds[kd * C + c];
                float a[4], bb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = bias;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float w = wd[r];
                    a[0] = fmaf(w, dq[r].x, a[0]); a[1] = fmaf(w, dq[r].y, a[1]);
                    a[2] = fmaf(w, dq[r].z, a[2]); a[3] = fmaf(w, dq[r].w, a[3]);
                }
                float P = 1.f, S = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? 3 - i : i;
                    // softplus and exp(dt * A) share the base-2 logarithm: lg = log2(1 + e^z) (z log2 e beyond the threshold),
                    // dt = lg ln 2, exp(dt A) = 2^(A lg)
                    const float z = a[e];
                    const float lg = z <= 20.f ? __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(z * 1.44269504088896340736f)) : z * 1.44269504088896340736f;
                    const float dl = lg * 0.69314718055994530942f;
                    a[e] = __builtin_amdgcn_exp2f(lg * Ak);
                    bb[e] = dl * Bv[e] * xv[e];
                    S = fmaf(a[e], S, bb[e]);
                    P = P * a[e];
                }
                float Pe, Se;
                if (dir) wave_scan_affine<true>(P, S, Pe, Se);
                else wave_scan_affine<false>(P, S, Pe, Se);
                float hw = carry[ch];
                if (NW > 1) {
                    float* ag = agg[slot];
                    if (lane == (dir ? 0 : BEM_WAVE - 1)) { ag[2 * wave] = P; ag[2 * wave + 1] = S; }
                    __syncthreads();
                    // compose the NW wave aggregates in scan order with a 16-lane DPP row scan (lane l < NW holds wave l of
                    // that order) instead of every thread walking all NW pairs: ~30 instructions instead of 5 NW
                    static_assert(NW <= 16, "the cross-wave scan uses one DPP row");
                    const int sl = min(lane, NW - 1), src = dir ? NW - 1 - sl : sl;      // always a valid pair; lanes >= NW hold the identity
                    const float Pl = ag[2 * src], Sl = ag[2 * src + 1];
                    float Pw = lane < NW ? Pl : 1.f, Sw = lane < NW ? Sl : 0.f;
#define BEM_ROW_STEP(DPP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " DPP "\n\tv_mul_f32_dpp %1, %1, %1 " DPP : "+v"(Sw), "+v"(Pw))
                    BEM_ROW_STEP("row_shr:1 row_mask:0xf bank_mask:0xf"); BEM_ROW_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
                    if (NW > 4) BEM_ROW_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
                    if (NW > 8) BEM_ROW_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
#undef BEM_ROW_STEP
                    const int rw = dir ? NW - 1 - wave : wave;                 // this wave's place in scan order (uniform)
                    const float Pt = lane_bcast(Pw, NW - 1), St = lane_bcast(Sw, NW - 1);
                    const float Px = lane_bcast(Pw, rw > 0 ? rw - 1 : 0), Sx = lane_bcast(Sw, rw > 0 ? rw - 1 : 0);
                    const float c0 = carry[ch];
                    hw = rw > 0 ? fmaf(Px, c0, Sx) : c0;
                    carry[ch] = fmaf(Pt, c0, St);
                    // the slot is rewritten NSLOT channel steps later; the barriers of the steps in between order that
                    // write after every read above
                    slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
                } else {
                    const float Pt = lane_bcast(P, dir ? 0 : BEM_WAVE - 1), St = lane_bcast(S, dir ? 0 : BEM_WAVE - 1);
                    carry[ch] = fmaf(Pt, hw, St);
                }
                float hh = fmaf(Pe, hw, Se);
                if (LATE_C) {
                    const float4 cq = *reinterpret_cast<const float4*>(xd + (int64_t)(R + 1) * L + pos);
                    Cv[0] = cq.x; Cv[1] = cq.y; Cv[2] = cq.z; Cv[3] = cq.w;
                }
                float yv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = dir ? 3 - i : i;
                    hh = fmaf(a[e], hh, bb[e]);
                    yv[e] = fmaf(Cv[e], hh, Dk * xv[e]);
                }
                if (ORI == 1 && TR) {
                    // the orientation-1-only launch of the row-major form keeps nothing in registers across the two directions
                    // (64-VGPR budget): the first direction parks its result in this workgroup's own output plane in SCAN order
                    // (coalesced, L2), the second adds it and drops the sum into the LDS slots of the x values it has just
                    // consumed (each slot is read by its owner thread only); the write-out below then overwrites the plane.
                    float* park = yb + (int64_t)c * L + pos;
                    if (dir == 0) {
                        *reinterpret_cast<float4*>(park) = make_float4(yv[0], yv[1], yv[2], yv[3]);
                    } else {
                        const float4 y0v = *reinterpret_cast<const float4*>(park);
                        float* pl = plane_sm + ch * psz + lofs;
                        pl[0] = y0v.x + yv[0]; pl[pitch] = y0v.y + yv[1]; pl[2 * pitch] = y0v.z + yv[2]; pl[3 * pitch] = y0v.w + yv[3];
                    }
                } else {
                    if (dir == 0) y[ch][k] = make_float4(yv[0], yv[1], yv[2], yv[3]);
                    else { y[ch][k].x += yv[0]; y[ch][k].y += yv[1]; y[ch][k].z += yv[2]; y[ch][k].w += yv[3]; }
                    // pin the result here: otherwise the optimiser sinks every step's replay down to the final stores and keeps
                    // each step's coefficients alive until then (~50 registers per tile)
                    asm volatile("" : "+v"(y[ch][k].x), "+v"(y[ch][k].y), "+v"(y[ch][k].z), "+v"(y[ch][k].w));
                }
            }
        }
    }
    if (via_lds) {
        __syncthreads();                                 // every read of the staged x planes is done
#pragma unroll
        for (int ch = 0; ch < CB; ++ch)
#pragma unroll
            for (int k = 0; k < (ORI == 1 ? 0 : T); ++k) {
                const int pos = (k * NT + threadIdx.x) * 4, pcol = pos / Himg, prow = pos - pcol * Himg;
                float* pl = plane_sm + ch * psz + (prow >> 2) * gpitch + pcol;
                pl[0] = y[ch][k].x; pl[pitch] = y[ch][k].y; pl[2 * pitch] = y[ch][k].z; pl[3 * pitch] = y[ch][k].w;
            }
        __syncthreads();
#pragma unroll
        for (int ch = 0; ch < CB; ++ch) {
            const int c = g * CB + ch;
            if (c < C) {
                const float* pl = plane_sm + ch * psz;
                float* dst = yb + (int64_t)c * L;
                for (int i = threadIdx.x * 4; i < L; i += NT * 4) {
                    const int row = i / Wimg, col = i - row * Wimg;
                    const float* sp = pl + (row >> 2) * gpitch + (row & 3) * pitch + col;
                    *reinterpret_cast<float4*>(dst + i) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ch = 0; ch < CB; ++ch) {
        const int c = g * CB + ch;
        if (c < C) {
#pragma unroll
            for (int k = 0; k < T; ++k) *reinterpret_cast<float4*>(yb + (int64_t)c * L + (k * NT + threadIdx.x) * 4) = y[ch][k];
        }
    }
}

// Chunked form without bounds masks for L % (NT * E) == 0 (L = 16384: 8 chunks of 2048): same chunk / carry structure as
// ss2d_scan_kernel, every load unconditional so the loads of a chunk issue back to back.
template <int NT, int E, int RT>
__global__ __launch_bounds__(NT) void ss2d_scan_chunks_kernel(
    const float* __restrict__ x0, const float* __restrict__ x1, const float* __restrict__ xd0,
    const float* __restrict__ xd1, const float* __restrict__ dtw, const float* __restrict__ dtb,
    const float* __restrict__ A, const float* __restrict__ Ds, float* __restrict__ y0,
    float* __restrict__ y1, int Bn, int C, int L, int Rr, int64_t xbs0, int64_t xbs1) {
    __shared__ float agg[2 * (NT / BEM_WAVE)];
    constexpr int CH = NT * E;
    const int R = RT ? RT : Rr;
    const int total = gridDim.x, lin = blockIdx.x;      // XCD-aware order, as in ss2d_scan_kernel
    const int per = total / 8, rem = total % 8, xcd = lin % 8, idx = lin / 8;
    const int wi = xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
    const int c = wi % C, b = (wi / C) % Bn, o = wi / (C * Bn);
    const float* xr = (o ? x1 : x0) + ((int64_t)b * C + c) * L;
    const float* xd = (o ? xd1 + (int64_t)b * xbs1 : xd0 + (int64_t)b * xbs0);
    const float* xdr = xd + (int64_t)(R + 2) * L;
    float* yr = (o ? y1 : y0) + ((int64_t)b * C + c) * L;
    const int kf = o, kr = o + 2;
    const float* wf = dtw + ((int64_t)kf * C + c) * R;
    const float* wr = dtw + ((int64_t)kr * C + c) * R;
    const float bf = dtb[kf * C + c], br = dtb[kr * C + c];
    const float Af = A[kf * C + c], Ar = A[kr * C + c];
    const float Df = Ds[kf * C + c], Dr = Ds[kr * C + c];
    const int nchunks = L / CH;
    float carry = 0.f;
    for (int j = 0; j < nchunks; ++j) {
        const int t0 = j * CH + threadIdx.x * E;
        float x[E], a[E], bb[E], y[E];
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(xr + t0 + i);
            x[i] = q.x; x[i + 1] = q.y; x[i + 2] = q.z; x[i + 3] = q.w;
        }
        float4 cq[E / 4];
#pragma unroll
        for (int i = 0; i < E; i += 4) cq[i / 4] = *reinterpret_cast<const float4*>(xd + (int64_t)(R + 1) * L + t0 + i);
        ss2d_coeffs_full<E, RT>(xd, wf, bf, Af, x, t0, L, R, a, bb);
        float hh = block_scan_enter<NT, E, false>(a, bb, carry, agg);
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float cv[4] = {cq[i / 4].x, cq[i / 4].y, cq[i / 4].z, cq[i / 4].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hh = a[i + k] * hh + bb[i + k];
                y[i + k] = fmaf(cv[k], hh, Df * x[i + k]);
            }
        }
#pragma unroll
        for (int i = 0; i < E; i += 4)
            *reinterpret_cast<float4*>(yr + t0 + i) = make_float4(y[i], y[i + 1], y[i + 2], y[i + 3]);
    }
    carry = 0.f;
    for (int j = nchunks - 1; j >= 0; --j) {
        const int t0 = j * CH + threadIdx.x * E;
        float x[E], a[E], bb[E], y[E];
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(xr + t0 + i);
            x[i] = q.x; x[i + 1] = q.y; x[i + 2] = q.z; x[i + 3] = q.w;
        }
        float4 cq[E / 4], yq[E / 4];
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            cq[i / 4] = *reinterpret_cast<const float4*>(xdr + (int64_t)(R + 1) * L + t0 + i);
            yq[i / 4] = *reinterpret_cast<const float4*>(yr + t0 + i);       // this thread's own earlier stores
        }
        ss2d_coeffs_full<E, RT>(xdr, wr, br, Ar, x, t0, L, R, a, bb);
        float hh = block_scan_enter<NT, E, true>(a, bb, carry, agg);
#pragma unroll
        for (int i = E - 4; i >= 0; i -= 4) {
            const float cv[4] = {cq[i / 4].x, cq[i / 4].y, cq[i / 4].z, cq[i / 4].w};
            const float yv[4] = {yq[i / 4].x, yq[i / 4].y, yq[i / 4].z, yq[i / 4].w};
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                hh = a[i + k] * hh + bb[i + k];
                y[i + k] = yv[k] + fmaf(cv[k], hh, Dr * x[i + k]);
            }
        }
#pragma unroll
        for (int i = 0; i < E; i += 4)
            *reinterpret_cast<float4*>(yr + t0 + i) = make_float4(y[i], y[i + 1], y[i + 2], y[i + 3]);
    }
}

// ------------------------------------------------------------------------------------------------
// cross scan / merge (operator seam only; the fused path never materialises the 4x tensor)
// ------------------------------------------------------------------------------------------------
__global__ void cross_scan_kernel(const float* __restrict__ x, float* __restrict__ xs, int C, int H, int W) {
    // grid: (ceil(L/256), C, B); xs (B,4,C,L)
    const int L = H * W;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const float* xp = x + ((int64_t)b * C + c) * L;
    float* o = xs + (((int64_t)b * 4) * C + c) * L;
    const int64_t ks = (int64_t)C * L;
    const float v0 = xp[l];
    const int wq = l / H, hq = l % H;            // column-major index l -> pixel (hq, wq)
    const float v1 = xp[hq * W + wq];
    o[l] = v0;
    o[ks + l] = v1;
    o[2 * ks + (L - 1 - l)] = v0;
    o[3 * ks + (L - 1 - l)] = v1;
}

__global__ void cross_merge_kernel(const float* __restrict__ ys, float* __restrict__ y, int C, int H, int W) {
    const int L = H * W;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int c = blockIdx.y, b = blockIdx.z;
    const float* p = ys + (((int64_t)b * 4) * C + c) * L;
    const int64_t ks = (int64_t)C * L;
    const int h = l / W, w = l % W;
    const int lc = w * H + h;
    y[((int64_t)b * C + c) * L + l] = (p[l] + p[2 * ks + (L - 1 - l)]) + (p[ks + lc] + p[3 * ks + (L - 1 - lc)]);
}

}  // namespace

template <typename TI>
static int selective_scan_fwd_launch(const TI* u, const TI* delta, const float* A, const TI* Bm, const TI* Cm, const float* D, const float* delta_bias,
                                     float* out, int batch, int dim, int L, int dstate, int ngroups, int delta_softplus, void* stream) {
    BEM_REQUIRE(u && delta && A && Bm && Cm && out, "selective_scan_fwd: null tensor");
    BEM_REQUIRE(batch >= 0 && dim > 0 && L >= 0, "selective_scan_fwd: bad shape (%d,%d,%d)", batch, dim, L);
    BEM_REQUIRE(dstate >= 1 && dstate <= 256, "selective_scan_fwd: dstate %d not in [1,256]", dstate);
    BEM_REQUIRE(ngroups >= 1 && dim % ngroups == 0, "selective_scan_fwd: dim %d %% ngroups %d != 0", dim, ngroups);
    BEM_REQUIRE(batch <= 65535, "selective_scan_fwd: batch %d > 65535", batch);
    BEM_REQUIRE(sizeof(TI) == 4 || L % 4 != 0 || (((uintptr_t)u | (uintptr_t)delta | (uintptr_t)Bm | (uintptr_t)Cm) & 7) == 0,
                "selective_scan_fwd: 16-bit inputs must be 8-byte aligned");
    if (batch == 0 || L == 0) return BEM_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(dim, batch);
    if (L <= 256)
        selective_scan_fwd_kernel<64, 4, TI><<<grid, 64, 0, s>>>(u, delta, A, Bm, Cm, D, delta_bias, out, dim, L, dstate, ngroups, delta_softplus);
    else if (L <= 1024)
        selective_scan_fwd_kernel<128, 8, TI><<<grid, 128, 0, s>>>(u, delta, A, Bm, Cm, D, delta_bias, out, dim, L, dstate, ngroups, delta_softplus);
    else
        selective_scan_fwd_kernel<256, 8, TI><<<grid, 256, 0, s>>>(u, delta, A, Bm, Cm, D, delta_bias, out, dim, L, dstate, ngroups, delta_softplus);
    return bem_check_launch("selective_scan_fwd");
}

extern "C" int bem_selective_scan_fwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                                          const float* Cm, const float* D, const float* delta_bias, float* out,
                                          int batch, int dim, int L, int dstate, int ngroups, int delta_softplus,
                                          void* stream) {
    return selective_scan_fwd_launch<float>(u, delta, A, Bm, Cm, D, delta_bias, out, batch, dim, L, dstate, ngroups, delta_softplus, stream);
}

// u, delta, B, C in float16 (in_dtype 1) or bfloat16 (2), read as they are; A, D, delta_bias and the output float32 (the reference's
// out_float = true form; selective_scan_oflex.cpp:166-216)
extern "C" int bem_selective_scan_fwd_in16(const void* u, const void* delta, const float* A, const void* Bm, const void* Cm, const float* D,
                                           const float* delta_bias, float* out, int in_dtype, int batch, int dim, int L, int dstate, int ngroups,
                                           int delta_softplus, void* stream) {
    BEM_REQUIRE(in_dtype == 1 || in_dtype == 2, "selective_scan_fwd_in16: in_dtype %d (1 = float16, 2 = bfloat16)", in_dtype);
    if (in_dtype == 1)
        return selective_scan_fwd_launch<bem_half_t>((const bem_half_t*)u, (const bem_half_t*)delta, A, (const bem_half_t*)Bm, (const bem_half_t*)Cm, D,
                                                     delta_bias, out, batch, dim, L, dstate, ngroups, delta_softplus, stream);
    return selective_scan_fwd_launch<bem_bf16_t>((const bem_bf16_t*)u, (const bem_bf16_t*)delta, A, (const bem_bf16_t*)Bm, (const bem_bf16_t*)Cm, D,
                                                 delta_bias, out, batch, dim, L, dstate, ngroups, delta_softplus, stream);
}

extern "C" int bem_ss2d_scan_strided_f32(const float* x0, const float* x1, const float* xd0, const float* xd1,
                                         const float* dtw, const float* dtb, const float* A, const float* Ds, float* y0,
                                         float* y1, int B, int C, int L, int R, int64_t xd0_bstride, int64_t xd1_bstride, void* stream);

extern "C" int bem_ss2d_scan_f32(const float* x0, const float* x1, const float* xd0, const float* xd1,
                                 const float* dtw, const float* dtb, const float* A, const float* Ds, float* y0,
                                 float* y1, int B, int C, int L, int R, void* stream) {
    return bem_ss2d_scan_strided_f32(x0, x1, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, L, R, 0, 0, stream);
}

extern "C" int bem_ss2d_scan_strided_f32(const float* x0, const float* x1, const float* xd0, const float* xd1,
                                         const float* dtw, const float* dtb, const float* A, const float* Ds, float* y0,
                                         float* y1, int B, int C, int L, int R, int64_t xd0_bstride, int64_t xd1_bstride, void* stream) {
    const int64_t xbs0 = xd0_bstride ? xd0_bstride : (int64_t)2 * (R + 2) * L, xbs1 = xd1_bstride ? xd1_bstride : (int64_t)2 * (R + 2) * L;
    BEM_REQUIRE(xbs0 >= (int64_t)2 * (R + 2) * L && xbs1 >= (int64_t)2 * (R + 2) * L && (L % 4 != 0 || (xbs0 % 4 == 0 && xbs1 % 4 == 0)), "ss2d_scan: x_dbl batch strides");
    BEM_REQUIRE(x0 && x1 && xd0 && xd1 && dtw && dtb && A && Ds && y0 && y1, "ss2d_scan: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && L >= 0 && R >= 1 && (int64_t)B * C * 2 < (1ll << 31), "ss2d_scan: bad shape B=%d C=%d L=%d R=%d", B, C, L, R);
    if (B == 0 || L == 0) return BEM_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = C * B * 2;
#define BEM_SS2D(NT, E) ss2d_scan_kernel<NT, E><<<grid, NT, 0, s>>>(x0, x1, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, L, R, xbs0, xbs1)
#define BEM_SS2D_FULL_R(NT, E, RT) ss2d_scan_full_kernel<NT, E, RT><<<grid, NT, 0, s>>>(x0, x1, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, R, xbs0, xbs1)
#define BEM_SS2D_FULL(NT, E) BEM_SS2D_FULL_R(NT, E, 0)      // runtime dt_rank: unrolling its plane loads measured slower (register pressure)
#define BEM_SS2D_CHUNKS(RT) ss2d_scan_chunks_kernel<256, 8, RT><<<grid, 256, 0, s>>>(x0, x1, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, L, R, xbs0, xbs1)
    const bool al = (((uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)xd0 | (uintptr_t)xd1 | (uintptr_t)y0 | (uintptr_t)y1) & 15) == 0;
    // channel-blocked whole-row forms for the plane sizes and dt_ranks of a 256x256 image (n_feat 40: C = 40 / 80 / 160)
#define BEM_SS2D_ROWS(NT, T, CB, RT, MW) do { ss2d_scan_rows_kernel<NT, T, CB, RT, MW, false><<<((C + CB - 1) / CB) * B * 2, NT, 0, s>>>( \
        x0, x1, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, xbs0, xbs1, 0); return bem_check_launch("ss2d_scan"); } while (0)
    // the tilings that won the round-1 / round-2 sweeps (the other 14 instantiations went with their A/B switch in round 3)
    if (al) {
        if (L == 1024 && R == 10) BEM_SS2D_ROWS(256, 1, 4, 10, 5);
        if (L == 4096 && R == 5) BEM_SS2D_ROWS(512, 2, 2, 5, 6);       // 139 us (256 x 4 tiles x 4 channels: 157)
        if (L == 16384 && R == 3) BEM_SS2D_ROWS(1024, 4, 1, 3, 8);     // 64 VGPRs: two 1024-thread workgroups per CU
    }
#undef BEM_SS2D_ROWS
    // whole-row forms: x read once, y written once.  (A 1024 x 16 form for L = 16384 measured slower than the chunked
    // kernel -- 128-VGPR budget, 2 workgroups per CU -- so rows longer than 4096 stay on the chunked path.)
    if (al && L == 4096) BEM_SS2D_FULL(512, 8);
    else if (al && L == 1024) BEM_SS2D_FULL(128, 8);
    else if (al && L == 256) BEM_SS2D_FULL(64, 4);
    else if (al && L > 2048 && L % 2048 == 0) {
        if (R == 3) BEM_SS2D_CHUNKS(3); else BEM_SS2D_CHUNKS(0);
    }
    else if (L <= 256) BEM_SS2D(64, 4);
    else if (L <= 1024) BEM_SS2D(128, 8);
    else BEM_SS2D(256, 8);
#undef BEM_SS2D
#undef BEM_SS2D_FULL
#undef BEM_SS2D_FULL_R
#undef BEM_SS2D_CHUNKS
    return bem_check_launch("ss2d_scan");
}

extern "C" int bem_cross_scan_f32(const float* x, float* xs, int B, int C, int H, int W, void* stream) {
    BEM_REQUIRE(x && xs, "cross_scan: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && C > 0 && C <= 65535 && H > 0 && W > 0, "cross_scan: bad shape");
    if (B == 0) return BEM_OK;
    dim3 grid(cdiv(H * W, 256), C, B);
    cross_scan_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, xs, C, H, W);
    return bem_check_launch("cross_scan");
}

extern "C" int bem_cross_merge_f32(const float* ys, float* y, int B, int C, int H, int W, void* stream) {
    BEM_REQUIRE(ys && y, "cross_merge: null tensor");
    BEM_REQUIRE(B >= 0 && B <= 65535 && C > 0 && C <= 65535 && H > 0 && W > 0, "cross_merge: bad shape");
    if (B == 0) return BEM_OK;
    dim3 grid(cdiv(H * W, 256), C, B);
    cross_merge_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(ys, y, C, H, W);
    return bem_check_launch("cross_merge");
}

extern "C" int64_t bem_selective_scan_bwd_ws_elems(int batch, int dim, int L, int dstate) {
    return (int64_t)batch * dim * cdiv(L > 0 ? L : 1, 1024) * dstate;
}

extern "C" int bem_selective_scan_bwd_f32(const float* u, const float* delta, const float* A, const float* Bm,
                                          const float* Cm, const float* D, const float* delta_bias, const float* dout,
                                          float* ws, float* du, float* ddelta, float* dA, float* dB, float* dC, float* dD,
                                          float* ddelta_bias, int batch, int dim, int L, int dstate, int ngroups,
                                          int delta_softplus, void* stream) {
    BEM_REQUIRE(u && delta && A && Bm && Cm && dout && ws && du && ddelta && dA && dB && dC, "selective_scan_bwd: null tensor");
    BEM_REQUIRE((D == nullptr) == (dD == nullptr) && (delta_bias == nullptr) == (ddelta_bias == nullptr),
                "selective_scan_bwd: dD / ddelta_bias must be given exactly when D / delta_bias are");
    BEM_REQUIRE(batch >= 0 && batch <= 65535 && dim > 0 && L >= 0, "selective_scan_bwd: bad shape");
    BEM_REQUIRE(dstate >= 1 && dstate <= 256 && ngroups >= 1 && dim % ngroups == 0, "selective_scan_bwd: dstate / groups");
    if (batch == 0 || L == 0) return BEM_OK;
    hipStream_t s = (hipStream_t)stream;
    // accumulated outputs start from zero (the reference allocates them with zeros_like, selective_scan_oflex.cpp:318-332)
    bool ok = hipMemsetAsync(dA, 0, sizeof(float) * (size_t)dim * dstate, s) == hipSuccess;
    ok = ok && hipMemsetAsync(dB, 0, sizeof(float) * (size_t)batch * ngroups * dstate * L, s) == hipSuccess;
    ok = ok && hipMemsetAsync(dC, 0, sizeof(float) * (size_t)batch * ngroups * dstate * L, s) == hipSuccess;
    if (dD) ok = ok && hipMemsetAsync(dD, 0, sizeof(float) * (size_t)dim, s) == hipSuccess;
    if (ddelta_bias) ok = ok && hipMemsetAsync(ddelta_bias, 0, sizeof(float) * (size_t)dim, s) == hipSuccess;
    if (!ok) return bem_check_launch("selective_scan_bwd memset");
    dim3 grid(dim, batch);
    selective_scan_bwd_kernel<256, 4><<<grid, 256, 0, s>>>(u, delta, A, Bm, Cm, D, delta_bias, dout, ws, du, ddelta, dA, dB, dC, dD,
                                                           ddelta_bias, dim, L, dstate, ngroups, delta_softplus);
    return bem_check_launch("selective_scan_bwd");
}

// Row-major in / row-major out form of the fused SS2D scan: x (B,C,H,W) serves both orientations, y0 and y1 are both
// (B,C,H,W) row-major (y1 = directions 1 + 3, already transposed back).  Only the plane sizes of a 256x256 image with
// their dt_ranks (L = 16384 / R = 3, 4096 / 5, 1024 / 10; H, W multiples of 4) -- bem_ss2d_scan_rm_supported tells.
extern "C" int bem_ss2d_scan_rm_supported(int H, int W, int R) {
    const int64_t L = (int64_t)H * W;
    return H % 4 == 0 && W % 4 == 0 && ((L == 16384 && R == 3) || (L == 4096 && R == 5) || (L == 1024 && R == 10));
}

// ORI = -1: one launch, both orientations (the orientation-1 workgroups stage through LDS).  ORI = 0 / 1: a launch for one
// orientation only -- the LDS path then is a compile-time property of the kernel, its result of the first direction is parked in
// the workgroup's own output plane instead of 16 registers, and neither variant needs scratch at the 64-register budget
// (the combined L = 16384 kernel spilled 44 bytes per lane: 1.6x its output bytes in HBM writes).
template <int NT, int T, int CB, int RT, int MW, int ORI>
static int launch_rows_tr(const float* x, const float* xd0, const float* xd1, const float* dtw, const float* dtb, const float* A,
                          const float* Ds, float* y0, float* y1, int B, int C, int H, int W, int64_t xbs0, int64_t xbs1, hipStream_t s) {
    const size_t lds = ORI == 0 ? 0 : (size_t)CB * (H / 4) * (4 * (W + 1) + 1) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ss2d_scan_rows_kernel<NT, T, CB, RT, MW, true, ORI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
        attr_set = true;
    }
    const int groups = ((C + CB - 1) / CB) * B;
    ss2d_scan_rows_kernel<NT, T, CB, RT, MW, true, ORI><<<ORI < 0 ? 2 * groups : groups, NT, lds, s>>>(x, x, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C,
                                                                                                        xbs0, xbs1, H);
    return bem_check_launch("ss2d_scan_rm");
}

extern "C" int bem_ss2d_scan_rm_f32(const float* x, const float* xd0, const float* xd1, const float* dtw, const float* dtb,
                                    const float* A, const float* Ds, float* y0, float* y1, int B, int C, int H, int W, int R,
                                    int64_t xd0_bstride, int64_t xd1_bstride, void* stream) {
    BEM_REQUIRE(x && xd0 && xd1 && dtw && dtb && A && Ds && y0 && y1, "ss2d_scan_rm: null tensor");
    BEM_REQUIRE(B >= 0 && C > 0 && H > 0 && W > 0 && R >= 1 && (int64_t)B * C * 2 < (1ll << 31), "ss2d_scan_rm: bad shape");
    BEM_REQUIRE(bem_ss2d_scan_rm_supported(H, W, R), "ss2d_scan_rm: unsupported plane %dx%d / dt_rank %d", H, W, R);
    const int L = H * W;
    const int64_t xbs0 = xd0_bstride ? xd0_bstride : (int64_t)2 * (R + 2) * L, xbs1 = xd1_bstride ? xd1_bstride : (int64_t)2 * (R + 2) * L;
    BEM_REQUIRE(xbs0 >= (int64_t)2 * (R + 2) * L && xbs1 >= (int64_t)2 * (R + 2) * L && xbs0 % 4 == 0 && xbs1 % 4 == 0, "ss2d_scan_rm: x_dbl batch strides");
    BEM_REQUIRE((((uintptr_t)x | (uintptr_t)xd0 | (uintptr_t)xd1 | (uintptr_t)y0 | (uintptr_t)y1) & 15) == 0, "ss2d_scan_rm: 16-byte alignment");
    if (B == 0) return BEM_OK;
    hipStream_t s = (hipStream_t)stream;
#define BEM_TR(NT, T, CB, RT, MW, ORI) launch_rows_tr<NT, T, CB, RT, MW, ORI>(x, xd0, xd1, dtw, dtb, A, Ds, y0, y1, B, C, H, W, xbs0, xbs1, s)
    if (L == 16384) {
        // two single-orientation launches: 2 x 172 us, no scratch; the combined kernel needs both orientations' state at once and spilled (372 us)
        const int rc = BEM_TR(1024, 4, 1, 3, 8, 0);
        return rc ? rc : BEM_TR(1024, 4, 1, 3, 8, 1);
    }
    if (L == 4096) return BEM_TR(512, 2, 2, 5, 6, -1);       // combined: 143 us against 152 us for the best split
    return BEM_TR(256, 1, 4, 10, 5, -1);
#undef BEM_TR
}
