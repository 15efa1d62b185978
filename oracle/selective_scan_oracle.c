/* CPU oracle (test infrastructure only; never linked into the product library).
 *
 * Plain-C restatement of the reference's selective scan semantics
 *   basicsr/vmamba/models/csms6s.py:29-72            (selective_scan_torch, the reference's CPU path)
 *   kernels/selective_scan/csrc/selective_scan/cusoflex/selective_scan_fwd_kernel_oflex.cuh:113-170
 * for float32 tensors:
 *   dt  = delta + delta_bias ; if softplus: dt = dt <= 20 ? log1p(exp(dt)) : dt
 *   h_n = exp(dt * A[d][n]) * h_n + dt * B[b][g][n][t] * u ;  y = sum_n C[b][g][n][t] * h_n + D[d] * u
 * with g = d / (dim / ngroups).  The recurrence is carried in float, step by step, like the
 * reference's Python loop.  Pinned by tests/golden/g1_*.npz (outputs of the reference itself).
 */
#include <math.h>
#include <stddef.h>

int oracle_selective_scan_f32(const float *u, const float *delta, const float *A, const float *Bm,
                              const float *Cm, const float *D, const float *delta_bias, float *out,
                              int batch, int dim, int L, int dstate, int ngroups, int delta_softplus)
{
    if (batch < 0 || dim <= 0 || L < 0 || dstate <= 0 || dstate > 256 || ngroups <= 0 || dim % ngroups) return 1;
    const int per = dim / ngroups;
    float h[256];
    for (int b = 0; b < batch; ++b)
        for (int d = 0; d < dim; ++d) {
            const int g = d / per;
            const float *ur = u + ((size_t)b * dim + d) * L;
            const float *dr = delta + ((size_t)b * dim + d) * L;
            float *yr = out + ((size_t)b * dim + d) * L;
            const float bias = delta_bias ? delta_bias[d] : 0.f;
            const float Dd = D ? D[d] : 0.f;
            for (int n = 0; n < dstate; ++n) h[n] = 0.f;
            for (int t = 0; t < L; ++t) {
                float dt = dr[t] + bias;
                if (delta_softplus) dt = dt <= 20.f ? log1pf(expf(dt)) : dt;
                const float uu = ur[t];
                float y = 0.f;
                for (int n = 0; n < dstate; ++n) {
                    const size_t bc = (((size_t)b * ngroups + g) * dstate + n) * L + t;
                    h[n] = expf(dt * A[(size_t)d * dstate + n]) * h[n] + dt * Bm[bc] * uu;
                    y += Cm[bc] * h[n];
                }
                yr[t] = y + Dd * uu;
            }
        }
    return 0;
}
