// Block-wide affine-map scan primitives shared by the forward (scan.hip) and backward (scan_bwd.hip) selective-scan kernels.
#pragma once
#include "bem_common.h"

namespace {

template <bool REV>
__device__ __forceinline__ float shfl_prev(float v, int d) {
    return REV ? __shfl_down(v, d, BEM_WAVE) : __shfl_up(v, d, BEM_WAVE);
}

// Block-wide composition of per-thread affine maps.  On entry (a[e], b[e]) are the per-position
// coefficients h_t = a_t * h_prev + b_t of this thread's E positions (identity = (1, 0) for padding).
// On exit h[e] holds the state after position e; `carry` (state entering the chunk, uniform) is updated
// to the state leaving the chunk.  agg is LDS scratch of 2*NW floats; contains two barriers.
// block_scan_enter returns the state entering this thread's first position (in scan order) and advances `carry`.
template <int NT, int E, bool REV>
__device__ __forceinline__ float block_scan_enter(const float (&a)[E], const float (&b)[E], float& carry, float* agg) {
    constexpr int NW = NT / BEM_WAVE;
    const int lane = threadIdx.x & (BEM_WAVE - 1);
    const int wave = threadIdx.x / BEM_WAVE;
    const int rl = REV ? (BEM_WAVE - 1 - lane) : lane;   // logical lane in scan order
    float P = 1.f, S = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = REV ? (E - 1 - i) : i;
        S = a[e] * S + b[e];
        P = P * a[e];
    }
    // inclusive wave scan of (P, S): compose(prev, cur) = (Pp*Pc, Pc*Sp + Sc)
#pragma unroll
    for (int d = 1; d < BEM_WAVE; d <<= 1) {
        const float Pp = shfl_prev<REV>(P, d);
        const float Sp = shfl_prev<REV>(S, d);
        if (rl >= d) {
            S = P * Sp + S;
            P = P * Pp;
        }
    }
    float Pe = shfl_prev<REV>(P, 1);
    float Se = shfl_prev<REV>(S, 1);
    if (rl == 0) { Pe = 1.f; Se = 0.f; }
    if (NW > 1) {
        if (rl == BEM_WAVE - 1) { agg[2 * wave] = P; agg[2 * wave + 1] = S; }
        __syncthreads();
    }
    float hw = carry;      // state entering this wave
    float hend = carry;    // state leaving the chunk
    if (NW > 1) {
        const int rw = REV ? (NW - 1 - wave) : wave;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int w = REV ? (NW - 1 - i) : i;
            const float Pw = agg[2 * w], Sw = agg[2 * w + 1];
            hend = Pw * hend + Sw;
            if (i < rw) hw = Pw * hw + Sw;
        }
        __syncthreads();   // agg may be rewritten by the next call
    } else {
        // single wave: total = inclusive value of the last logical lane
        const float Pt = __shfl(P, REV ? 0 : BEM_WAVE - 1, BEM_WAVE);
        const float St = __shfl(S, REV ? 0 : BEM_WAVE - 1, BEM_WAVE);
        hend = Pt * carry + St;
    }
    carry = hend;
    return Pe * hw + Se;       // state entering this thread
}

template <int NT, int E, bool REV>
__device__ __forceinline__ void block_scan_affine(const float (&a)[E], const float (&b)[E], float (&h)[E],
                                                  float& carry, float* agg) {
    float hh = block_scan_enter<NT, E, REV>(a, b, carry, agg);
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int e = REV ? (E - 1 - i) : i;
        hh = a[e] * hh + b[e];
        h[e] = hh;
    }
}

template <int E>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int64_t t0, int L, bool vec, float (&v)[E]) {
    if (vec && t0 + E <= L) {
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(p + t0 + i);
            v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = (t0 + i < L) ? p[t0 + i] : 0.f;
    }
}
template <int E>
__device__ __forceinline__ void store_row(float* __restrict__ p, int64_t t0, int L, bool vec, const float (&v)[E]) {
    if (vec && t0 + E <= L) {
#pragma unroll
        for (int i = 0; i < E; i += 4)
            *reinterpret_cast<float4*>(p + t0 + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (t0 + i < L) p[t0 + i] = v[i];
    }
}

template <int NT>
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, BEM_WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NT / BEM_WAVE; ++w) s += sh[w];
    return s;
}

}  // namespace
