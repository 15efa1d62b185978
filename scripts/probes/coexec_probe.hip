// Probe: do matrix-core and vector instructions overlap on one SIMD of gfx950 (a) inside one wave's instruction stream, (b) between two
// waves of the SIMD?  Build: hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip ; run on the GPU box.
// One workgroup per CU (grid 256), NW waves; every wave runs `iters` rounds of its role:
//   M: 8 x v_mfma_f32_32x32x16_bf16 (two accumulators)          V: 48 x v_pk_fma_f32 (8 independent chains)
//   I: both in one stream, 1 MFMA : 6 VALU by sched_group_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// VK: what the vector rounds are made of -- 0: 48 v_pk_fma_f32, 1: 48 v_fma_f32 (inline asm: never re-packed), 2: 24 v_exp_f32 (8 cycles each)
template <int VK>
__device__ __forceinline__ void vround(f32x2 (&v)[8], const f32x2 m, const f32x2 c) {
    if (VK == 0) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
    } else if (VK == 1) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(m[0]), "v"(c[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][1]) : "v"(m[1]), "v"(c[1]));
            }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i][0]));
    }
}

template <int MODE, int VK>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int prio) {
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {}, a1 = {};
    bf16x8 x, w;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(0.001f * (threadIdx.x + i)); w[i] = (__bf16)(0.002f * (threadIdx.x ^ i)); }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{0.5f + i, 0.25f * threadIdx.x};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    // role of this wave
    const bool do_m = MODE == 0 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave < 4) || (MODE == 5 && (wave & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || ((MODE == 3 || MODE == 4) && wave >= 4) || (MODE == 5 && (wave & 1) == 1);
    if (prio && do_m && !do_v) __builtin_amdgcn_s_setprio(3);
    if (MODE == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, a1, 0, 0, 0);
            }
            if (VK == 0) {
                vround<0>(v, m, c);
            } else {
                // asm statements keep their program order: interleave by hand, 1 MFMA : 6 (or 3 exp) vector instructions
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);   // 6 VALU
            }
        }
    } else {
        if (do_m)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, w, a1, 0, 0, 0);
                }
            }
        if (do_v)
            for (int it = 0; it < iters; ++it) vround<VK>(v, m, c);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int VK = 0>
float run(float* out, int threads, int iters, int prio, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, VK><<<grid, threads>>>(out, iters, prio);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<MODE, VK><<<grid, threads>>>(out, iters, prio);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / 5;
}

int main() {
    float* out; hipMalloc(&out, 1024 * 512 * 4);
    const int it = 20000;
    const double cyc_m = 8.0 * 32, cyc_v = 48.0 * 4;   // per round, if alone
    printf("per round alone: MFMA %.0f cycles, VALU %.0f cycles; iters %d\n", cyc_m, cyc_v, it);
    for (int grid : {256, 512}) {
        printf("grid %d (%d workgroup(s) per CU)\n", grid, grid / 256);
        printf("  M only, 4 waves/WG (1 per SIMD):      %8.1f us\n", run<0>(out, 256, it, 0, grid));
        printf("  V only, 4 waves/WG:                   %8.1f us\n", run<1>(out, 256, it, 0, grid));
        printf("  M only, 8 waves/WG (2 per SIMD):      %8.1f us\n", run<0>(out, 512, it, 0, grid));
        printf("  V only, 8 waves/WG:                   %8.1f us\n", run<1>(out, 512, it, 0, grid));
        printf("  interleaved in one stream, 4 waves:   %8.1f us\n", run<2>(out, 256, it, 0, grid));
        printf("  interleaved in one stream, 8 waves:   %8.1f us\n", run<2>(out, 512, it, 0, grid));
        printf("  waves 0-3 M, waves 4-7 V:             %8.1f us\n", run<3>(out, 512, it, 0, grid));
        printf("  same, M waves at s_setprio 3:         %8.1f us\n", run<4>(out, 512, it, 1, grid));
        printf("  even waves M, odd waves V:            %8.1f us\n", run<5>(out, 512, it, 0, grid));
        printf("  same, M waves at s_setprio 3:         %8.1f us\n", run<5>(out, 512, it, 1, grid));
        printf("  [v_fma_f32]  V only 4 waves %8.1f | 8 waves %8.1f | waves 0-3 M + 4-7 V %8.1f | even M odd V %8.1f | same prio %8.1f\n",
               run<1, 1>(out, 256, it, 0, grid), run<1, 1>(out, 512, it, 0, grid), run<3, 1>(out, 512, it, 0, grid), run<5, 1>(out, 512, it, 0, grid), run<5, 1>(out, 512, it, 1, grid));
        printf("  [v_exp_f32]  V only 4 waves %8.1f | 8 waves %8.1f | waves 0-3 M + 4-7 V %8.1f | even M odd V %8.1f | same prio %8.1f\n",
               run<1, 2>(out, 256, it, 0, grid), run<1, 2>(out, 512, it, 0, grid), run<3, 2>(out, 512, it, 0, grid), run<5, 2>(out, 512, it, 0, grid), run<5, 2>(out, 512, it, 1, grid));
    }
    return 0;
}
