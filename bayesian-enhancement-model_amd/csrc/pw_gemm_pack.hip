// f32 matrix-core operand packing: natural (nsets, M, K) weights -> (nsets, MT, ceil(K/2), 64 lanes) pairs in the operand order of
// v_mfma_f32_32x32x2_f32 -- what the implicit-GEMM convolutions of conv.hip (conv2d_mfma*) and the conv weight-gradient kernels consume.
// (The pointwise GEMMs themselves run on the bf16-limb kernels of pw_gemm_x6.hip; their f32-MFMA predecessors were removed in round 3.)
#include "bem_common.h"

namespace {

__global__ void pack_pw_weight_kernel(const float* __restrict__ W, float* __restrict__ Wp, int M, int K, int MT, int KS) {
    // grid: (ceil(MT*KS*64 / 256), nsets)
    const int64_t per = (int64_t)MT * KS * 64;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per) return;
    const int lane = (int)(i & 63);
    const int64_t t = i >> 6;
    const int st = (int)(t % KS), mt = (int)(t / KS);
    const int row = mt * 32 + (lane & 31), col = 2 * st + (lane >> 5);
    const int set = blockIdx.y;
    Wp[(int64_t)set * per + i] = (row < M && col < K) ? W[((int64_t)set * M + row) * K + col] : 0.f;
}

}  // namespace


extern "C" int64_t bem_pw_packed_elems(int M, int K) { return (int64_t)cdiv(M, 32) * cdiv(K, 2) * 64; }

extern "C" int bem_pack_pw_weight_f32(const float* W, float* Wp, int nsets, int M, int K, void* stream) {
    BEM_REQUIRE(W && Wp, "pack_pw_weight: null tensor");
    BEM_REQUIRE(nsets >= 0 && nsets <= 65535 && M > 0 && K > 0, "pack_pw_weight: bad shape");
    if (nsets == 0) return BEM_OK;
    const int MT = cdiv(M, 32), KS = cdiv(K, 2);
    dim3 grid((unsigned)cdiv64((int64_t)MT * KS * 64, 256), nsets);
    pack_pw_weight_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(W, Wp, M, K, MT, KS);
    return bem_check_launch("pack_pw_weight");
}
