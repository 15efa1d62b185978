// Shared device helpers of the x6 (3-limb bf16) matrix-core kernels: operand typedefs, the exact limb split, the six-product MAC
// and the XCD-aware tile order.  Included by pw_gemm_x6.hip, gdmlp_x6.hip, conv_rows_x6.hip and ss2d_front_x6.hip; see the header comment of pw_gemm_x6.hip.
#pragma once
#include "bem_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __builtin_bit_cast(float, u); }

// Workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2).  Give every XCD a contiguous run of pixel
// tiles so that the chunks one L2 collects (and later writes back) for a plane are adjacent in memory.
__device__ __forceinline__ int xcd_tile(int x, int nx) {
    const int per = nx >> 3, rem = nx & 7, xcd = x & 7, idx = x >> 3;
    return xcd < rem ? xcd * (per + 1) + idx : rem * (per + 1) + (xcd - rem) * per + idx;
}

// exact 3-limb split of 8 values (the lane's 8 channels of one k-block and sub-tile) into three MFMA operands.
// Limbs are taken by round-to-nearest-even (v_cvt_pk_bf16_f32, two values per instruction): |m| <= 2^-8 |v|,
// |l| <= 2^-16 |v|, the residuals are exact in f32, and the limb products the GEMM drops (m*l, l*m, l*l) are
// <= 2^-23 |x w| with no preferred sign.  (Truncated limbs are twice as large and all carry the sign of v: the same
// six products then leave a biased 2^-21 error.)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 a = {v[2 * q], v[2 * q + 1]};
        const uint32_t uh = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bf16x2));
        const f32x2 r1 = {a[0] - bitsf(uh << 16), a[1] - bitsf(uh & 0xffff0000u)};
        const uint32_t um = __builtin_bit_cast(uint32_t, __builtin_convertvector(r1, bf16x2));
        const f32x2 r2 = {r1[0] - bitsf(um << 16), r1[1] - bitsf(um & 0xffff0000u)};
        h[q] = uh;
        m[q] = um;
        l[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r2, bf16x2));
    }
}

__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// The six limb products in two accumulators: `hi` takes xh*wh (16-bit significands, magnitude of the result), `lo` the
// five products that are <= 2^-8 of it.  Feeding the small products straight into the large accumulator loses their
// low bits in the MFMA's addend alignment -- always downwards, a bias of ~1e-8 per output that adds up coherently over a
// network; kept apart they are summed among their own size and joined to `hi` by one rounded f32 add in the epilogue.
__device__ __forceinline__ void mac6(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x16& hi, f32x16& lo) {
    lo = mfma16(w[0], x[2], lo);
    lo = mfma16(w[2], x[0], lo);
    lo = mfma16(w[1], x[1], lo);
    lo = mfma16(w[0], x[1], lo);
    lo = mfma16(w[1], x[0], lo);
    hi = mfma16(w[0], x[0], hi);
}

// one 1 KiB piece global -> LDS: lane l moves the 16 bytes at sbase + voff (voff = 16 l) to lds_dst + 16 l.  sbase is wave-uniform (an
// SGPR pair: no per-lane 64-bit address arithmetic), M0 carries the wave-uniform LDS byte address.  Not visible to the compiler's
// wait-count bookkeeping: the caller drains with s_waitcnt vmcnt(0) before the barrier that publishes the bytes.
__device__ __forceinline__ void glds16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }   // low half of a generic LDS pointer = LDS offset

}  // namespace
