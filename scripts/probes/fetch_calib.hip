// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo's kernels use (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate
// on a known byte count").  Each kernel streams the same 1 GiB buffer once (coalesced, one element per lane and step) with 4-, 8- or 16-byte
// loads, or fills it with 4-, 8- or 16-byte stores.  Run under `rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`; the factor of a
// width is bytes / (counter * 1024).   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ void read_k(const T* __restrict__ p, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        if constexpr (sizeof(T) == 4) s += v; else if constexpr (sizeof(T) == 8) s += v[0] + v[1]; else s += v[0] + v[1] + v[2] + v[3];
    }
    if (s == 12345.678f) out[0] = s;
}
template <typename T>
__global__ void write_k(T* __restrict__ p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v;
        if constexpr (sizeof(T) == 4) v = 1.f; else if constexpr (sizeof(T) == 8) v = T{1.f, 2.f}; else v = T{1.f, 2.f, 3.f, 4.f};
        p[i] = v;
    }
}
// the row pattern of the NCHW kernels: a half-wave reads 32 consecutive floats (128 B) of one channel plane, the planes 64 KiB apart
__global__ void read_rows_k(const float* __restrict__ p, float* out, size_t nplanes, size_t plane) {
    float s = 0.f;
    const int lane = threadIdx.x & 31, half = (threadIdx.x >> 5) & 1, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x * 4 + wave; t < plane / 32; t += (size_t)gridDim.x * 4)
        for (size_t c = half; c < nplanes; c += 2) s += p[c * plane + t * 32 + lane];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf; float* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    const int grid = 256 * 8, blk = 256;
    for (int rep = 0; rep < 2; ++rep) {
        read_k<float><<<grid, blk>>>((const float*)buf, out, bytes / 4);
        read_k<f2><<<grid, blk>>>((const f2*)buf, out, bytes / 8);
        read_k<f4><<<grid, blk>>>((const f4*)buf, out, bytes / 16);
        read_rows_k<<<grid, blk>>>((const float*)buf, out, bytes / 4 / 16384, 16384);
        write_k<float><<<grid, blk>>>((float*)buf, bytes / 4);
        write_k<f2><<<grid, blk>>>((f2*)buf, bytes / 8);
        write_k<f4><<<grid, blk>>>((f4*)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
