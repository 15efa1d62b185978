// Shared helpers for the gfx950 kernels of libbem_hip.so (wave64 only, no other targets).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/bem_hip.h"

#define BEM_WAVE 64

extern thread_local char bem_err_buf[512];

#define BEM_REQUIRE(cond, ...)                                              \
    do {                                                                    \
        if (!(cond)) {                                                      \
            snprintf(bem_err_buf, sizeof(bem_err_buf), __VA_ARGS__);        \
            return BEM_ERR_INVALID;                                         \
        }                                                                   \
    } while (0)

static inline int bem_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(bem_err_buf, sizeof(bem_err_buf), "%s: %s", what, hipGetErrorString(e));
        return BEM_ERR_LAUNCH;
    }
    return BEM_OK;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Hardware transcendental forms (v_exp_f32 / v_log_f32, ~1 ulp): the scan evaluates three of them per
// element and direction, and the libm-accurate versions made that kernel instruction-bound.
__device__ __forceinline__ float bem_fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float bem_flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float bem_softplus(float x) {
    // F.softplus, threshold 20 (csms6s.py:54, selective_scan_fwd_kernel_oflex.cuh:125).  log(1 + e^x) is
    // formed directly: for e^x < 2^-24 it returns 0 instead of e^x, an absolute error below 6e-8.
    return x <= 20.f ? bem_flog(1.f + bem_fexp(x)) : x;
}
// A real v_mov_b32: values that came back from memory (LDS or global) are copied once before packed-f32 arithmetic may pair them up.  On gfx950 a
// v_pk_*_f32 working in place on an LDS-returned register pair through op_sel read the pair's pre-load content in lanes 48..63
// a few times per 10^7 outputs (two workgroups per CU; waits correct) -- DESIGN.md section 6.4, scripts/isa_audit.py check 2.
__device__ __forceinline__ float valu_copy(float v) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float bem_silu(float x) { return x / (1.f + bem_fexp(-x)); }
__device__ __forceinline__ float bem_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on the hardware exp / rcp: ~12 instructions instead of
// libm's branchy erff.  Used where GELU sits between two matrix-core phases (fused gdMlp).
__device__ __forceinline__ float bem_erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.f - p * t * bem_fexp(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float bem_gelu_fast(float x) { return 0.5f * x * (1.f + bem_erf_fast(x * 0.70710678118654752440f)); }

// Philox4x32-10 counter-based generator + Box-Muller: one N(0,1) draw per (element index, stream id) under a 64-bit seed.
// Shared by the Bayesian weight sampler (elementwise.hip) and the fused sample-and-pack kernel (pw_gemm_x6.hip): the same
// (i, seed, stream) gives the same draw in both.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// Four N(0,1) draws of counter block j (elements 4j .. 4j+3): both Box-Muller outputs of the two uniform pairs.
__device__ __forceinline__ void philox_normal4(int64_t j, uint64_t seed, uint64_t stream_id, float (&z)[4]) {
    uint32_t c[4] = {(uint32_t)j, (uint32_t)((uint64_t)j >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
        const float u2 = (float)(c[2 * h + 1] >> 8) * (1.0f / 16777216.0f);        // [0, 1)
        const float r = sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.28318530717958647692f * u2, &sn, &cs);
        z[2 * h] = r * cs;
        z[2 * h + 1] = r * sn;
    }
}
// draw of element i = component i & 3 of block i >> 2 (one Philox call serves four consecutive elements)
__device__ __forceinline__ float philox_normal(int64_t i, uint64_t seed, uint64_t stream_id) {
    float z[4];
    philox_normal4(i >> 2, seed, stream_id, z);
    return z[i & 3];
}
