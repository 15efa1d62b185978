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
// 16-bit inputs of the operator seam (selective_scan_oflex.cpp:166-216 takes f16 / bf16 u, delta, B, C): the same row access, 2 bytes per
// element, converted in registers -- no f32 copy of the tensor is made.
struct bem_half_t { uint16_t b; };
struct bem_bf16_t { uint16_t b; };
__device__ __forceinline__ float bem_to_f32(bem_half_t h) { return (float)__builtin_bit_cast(_Float16, h.b); }
__device__ __forceinline__ float bem_to_f32(bem_bf16_t h) { return __builtin_bit_cast(float, (uint32_t)h.b << 16); }
template <int E, typename T16>
__device__ __forceinline__ void load_row(const T16* __restrict__ p, int64_t t0, int L, bool vec, float (&v)[E]) {
    if (vec && t0 + E <= L) {                       // vec: L % 4 == 0 and the tensor 8-byte aligned -> 8-byte pieces of 4 elements
#pragma unroll
        for (int i = 0; i < E; i += 4) {
            const uint2 q = *reinterpret_cast<const uint2*>(p + t0 + i);
            v[i] = bem_to_f32(T16{(uint16_t)(q.x & 0xffffu)}); v[i + 1] = bem_to_f32(T16{(uint16_t)(q.x >> 16)});
            v[i + 2] = bem_to_f32(T16{(uint16_t)(q.y & 0xffffu)}); v[i + 3] = bem_to_f32(T16{(uint16_t)(q.y >> 16)});
        }
    } else {
#pragma unroll
        for (int i = 0; i < E; ++i) v[i] = (t0 + i < L) ? bem_to_f32(p[t0 + i]) : 0.f;
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

template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                 CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// One step of the lane scan of affine maps h -> P h + S: (P, S) <- (P, S) o (P, S)[source lane], i.e. S = P S' + S, P = P P'.
// Both halves are ONE DPP instruction each: the DPP operand is the instruction's own src0, and a lane without a source lane (row
// boundary, masked row) is simply not written -- the identity of the composition.  The builtin form (update_dpp with old = 1 / 0,
// then fma / mul) costs six instructions per step (v_mov old, v_mov_dpp, twice, + fma + mul), a third of the whole scan kernel.
// s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write of that VGPR (gfx9 hazard; the compiler cannot see into
// the asm, and S / P are written by the instruction right before the block).
#define BEM_SCAN_STEP(DPP)                                                                                   \
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " DPP "\n\tv_mul_f32_dpp %1, %1, %1 " DPP : "+v"(S), "+v"(P))
// inclusive scan of the per-lane affine maps (P, S) in ascending (REV = false) / descending (REV = true) lane order;
// returns the exclusive map (Pe, Se) of every lane and leaves the wavefront total in lane 63 (0 for REV).
template <bool REV>
__device__ __forceinline__ void wave_scan_affine(float& P, float& S, float& Pe, float& Se) {
    if (!REV) {
        BEM_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf"); BEM_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
        BEM_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf"); BEM_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
        BEM_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf"); BEM_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf");
        asm volatile("s_nop 1");              // the compiler's own DPP moves below read P / S right behind the asm writes
        Pe = dpp_mov<0x138, 0xf>(1.f, P);   // wave_shr 1
        Se = dpp_mov<0x138, 0xf>(0.f, S);
    } else {
        BEM_SCAN_STEP("row_shl:1 row_mask:0xf bank_mask:0xf"); BEM_SCAN_STEP("row_shl:2 row_mask:0xf bank_mask:0xf");
        BEM_SCAN_STEP("row_shl:4 row_mask:0xf bank_mask:0xf"); BEM_SCAN_STEP("row_shl:8 row_mask:0xf bank_mask:0xf");
        const int lane = threadIdx.x & 63;
        {   // rows 0 / 2 append the suffix of rows 1 / 3 (their lane 16 / 48)
            const float P16 = lane_bcast(P, 16), S16 = lane_bcast(S, 16), P48 = lane_bcast(P, 48), S48 = lane_bcast(S, 48);
            const bool take = (lane & 16) == 0;
            const float Pp = take ? ((lane & 32) ? P48 : P16) : 1.f, Sp = take ? ((lane & 32) ? S48 : S16) : 0.f;
            S = fmaf(P, Sp, S);
            P = P * Pp;
        }
        {   // rows 0, 1 append the suffix of rows 2, 3 (lane 32)
            const float P32 = lane_bcast(P, 32), S32 = lane_bcast(S, 32);
            const bool take = lane < 32;
            const float Pp = take ? P32 : 1.f, Sp = take ? S32 : 0.f;
            S = fmaf(P, Sp, S);
            P = P * Pp;
        }
        Pe = dpp_mov<0x130, 0xf>(1.f, P);   // wave_shl 1
        Se = dpp_mov<0x130, 0xf>(0.f, S);
    }
}
#undef BEM_SCAN_STEP

// Cross-wave step of a block scan whose wavefront part ran on wave_scan_affine<REV>: (P, S) is the inclusive map of the lane,
// i.e. the wavefront total sits in lane 63 (lane 0 for REV).  The NW totals go through the LDS slot `ag` (2 NW floats, ONE
// barrier; the caller rotates slots so that a slot is rewritten only after later barriers) and are composed in scan order on
// one 16-lane DPP row.  Returns the state entering this wavefront and advances `carry` (state entering the block) to the state
// leaving it.
template <int NW, bool REV>
__device__ __forceinline__ float cross_wave_affine(float P, float S, float* ag, float& carry) {
    static_assert(NW <= 16, "the cross-wave scan uses one DPP row");
    const int lane = threadIdx.x & (BEM_WAVE - 1), wave = threadIdx.x / BEM_WAVE;
    if (NW == 1) {
        const float Pt = lane_bcast(P, REV ? 0 : BEM_WAVE - 1), St = lane_bcast(S, REV ? 0 : BEM_WAVE - 1);
        const float hw = carry;
        carry = fmaf(Pt, hw, St);
        return hw;
    }
    if (lane == (REV ? 0 : BEM_WAVE - 1)) { ag[2 * wave] = P; ag[2 * wave + 1] = S; }
    __syncthreads();
    const int sl = min(lane, NW - 1), src = REV ? NW - 1 - sl : sl;
    const float Pl = ag[2 * src], Sl = ag[2 * src + 1];
    float Pw = lane < NW ? Pl : 1.f, Sw = lane < NW ? Sl : 0.f;
#define BEM_ROW_STEP(DPP) asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " DPP "\n\tv_mul_f32_dpp %1, %1, %1 " DPP : "+v"(Sw), "+v"(Pw))
    BEM_ROW_STEP("row_shr:1 row_mask:0xf bank_mask:0xf"); BEM_ROW_STEP("row_shr:2 row_mask:0xf bank_mask:0xf");
    if (NW > 4) BEM_ROW_STEP("row_shr:4 row_mask:0xf bank_mask:0xf");
    if (NW > 8) BEM_ROW_STEP("row_shr:8 row_mask:0xf bank_mask:0xf");
#undef BEM_ROW_STEP
    const int rw = REV ? NW - 1 - wave : wave;
    const float Pt = lane_bcast(Pw, NW - 1), St = lane_bcast(Sw, NW - 1);
    const float Px = lane_bcast(Pw, rw > 0 ? rw - 1 : 0), Sx = lane_bcast(Sw, rw > 0 ? rw - 1 : 0);
    const float c0 = carry;
    carry = fmaf(Pt, c0, St);
    return rw > 0 ? fmaf(Px, c0, Sx) : c0;
}

}  // namespace
