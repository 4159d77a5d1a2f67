// Shared device helpers for the GE2E HIP kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ge2e {

constexpr int kWave = 64;  // CDNA wavefront; hard-coded, this library targets gfx950 only

// Launch-time problem description shared by every kernel.
struct Problem {
    const float* E;   // [B][N][M][D]
    const float* w;   // device scalar (s3:16); null -> w_imm
    const float* b;   // device scalar (s3:17); null -> b_imm
    float w_imm, b_imm;
    float* loss;      // [B]
    float* per;       // [B][N][M] or null
    float* dE;        // [B][N][M][D] or null (forward only)
    float* dw;        // [B] or null
    float* db;        // [B] or null
    float* cos_out;   // [B][N][M][N] or null (ge2e_cos_sim)
    float* ws;        // workspace
    int B, N, M, D;
    int variant;
    float eps_cos;    // cosine_similarity eps (1e-8)
    float eps;        // hp.general.small_err (1e-6)
    float log_eps;    // logf(eps), -inf when eps == 0
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
    return v;
}
// (value, index) arg-max; ties resolve to the lowest index (torch.max picks the first).
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(v, o, kWave);
        int oi = __shfl_xor(i, o, kWave);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// x / max(|x|, eps) bookkeeping shared by every kernel: from a squared norm give
// the reciprocal of the clamped norm and kappa = clamped / true norm (0 for a zero
// vector).  ATen clamps the norm in place under no_grad, so the backward of
// x_hat = x / n_c is (g - kappa * (g . x_hat) * x_hat) / n_c  (oracle/_unit_bwd).
__device__ __forceinline__ void unit_stats(float sq, float eps_cos, float& rn, float& kappa) {
    float n = sqrtf(sq);
    float nc = fmaxf(n, eps_cos);
    rn = 1.0f / nc;
    kappa = n > 0.0f ? nc / n : 0.0f;
}

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace ge2e
