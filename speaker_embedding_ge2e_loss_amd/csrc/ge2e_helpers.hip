// The reference's static helpers are ordinary autograd code (s3:33-38, 41-80, 95-112, 114-127): callers may build a
// loss of their own from them and backpropagate.  This file holds their backward passes (and the forward of
// get_utterance_centroids) as plain fp32 kernels -- one wave per row, DPP reductions; they are eval / compatibility
// paths, not the training hot path (GE2ELoss.forward produces every gradient in its one fused launch).
//
// Conventions as everywhere: x-hat = x / max(|x|, eps_cos); its backward is (g - kappa (g . x-hat) x-hat) / n_c with
// kappa = clamped / true norm (0 for a zero vector), which is what ATen's cosine_similarity does (oracle/_unit_bwd).
#include "ge2e_common.hpp"
#include "ge2e_helpers.hpp"

namespace ge2e {

namespace {

// u_ji = (sum_i' e_ji' - e_ji) / (M - 1)  (s3:95-112).  The map is linear and symmetric: its backward is the same
// kernel applied to the incoming gradient.
__global__ __launch_bounds__(256) void utt_centroids_kernel(const float* E, int speakers, int M, int D, float* U) {
    const size_t total = (size_t)speakers * D;
    const float inv = 1.0f / (float)(M - 1);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t j = idx / D, d = idx % D;
        const float* e = E + j * M * D + d;
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += e[(size_t)i * D];
        float* u = U + j * M * D + d;
        for (int i = 0; i < M; ++i) u[(size_t)i * D] = (s - e[(size_t)i * D]) * inv;
    }
}

// backward of get_centroids (mean over the utterance axis): dE_ji = g_j / M
__global__ __launch_bounds__(256) void centroids_bwd_kernel(const float* g, int speakers, int M, int D, float* dE) {
    const size_t total = (size_t)speakers * M * D;
    const float inv = 1.0f / (float)M;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t j = idx / ((size_t)M * D), d = idx % D;
        dE[idx] = g[j * D + d] * inv;
    }
}

// ---- get_cos_sim backward, three small launches (the boundaries are the synchronisation) ---------------------------
// Local rows everywhere below (ge2e_generic.hip: ge2e_cos_centroids_kernel): E / dE hold the rows of the n speakers whose
// columns are j0 .. j0 + n - 1 of the N centroids in C; dC is then this shard's PARTIAL centroid gradient (the sharded
// loss sums it over the ranks with a reduce-scatter).  n = N, j0 = 0: the whole batch.
// K0: speaker sums S[b][jl][:] (block < n) and the centroid scalars CS[b][k] = (1 / max(|c_k|, eps), kappa_k) (block < N)
__global__ __launch_bounds__(64) void cos_bwd_k0(const float* E, const float* C, int n, int N, int M, int D, float eps_cos,
                                                 float* S, float* CS) {
    const int nb = n > N ? n : N;
    const int b = blockIdx.x / nb, j = blockIdx.x % nb, lane = threadIdx.x;
    if (j < n) {
        const float* e = E + ((size_t)b * n + j) * M * D;
        for (int d = lane; d < D; d += kWave) {
            float s = 0.f;
            for (int i = 0; i < M; ++i) s += e[(size_t)i * D + d];
            S[((size_t)b * n + j) * D + d] = s;
        }
    }
    if (j < N) {
        float sq = 0.f;
        for (int d = lane; d < D; d += kWave) {
            const float c = C[((size_t)b * N + j) * D + d];
            sq += c * c;
        }
        sq = wave_sum(sq);
        float rn, kap;
        unit_stats(sq, eps_cos, rn, kap);
        if (lane == 0) { CS[((size_t)b * N + j) * 2] = rn; CS[((size_t)b * N + j) * 2 + 1] = kap; }
    }
}

// K1: one wave per row r = (j, i): the a-slot part of dE_r and the leave-one-out slot vector du_r
//   g_e = sum_{k != j} g[r][k] c-hat_k + g[r][j] u-hat_r ;  dE_r = (g_e - kappa_e (g_e . e-hat) e-hat) / n_e
//   du_r = g[r][j] (e-hat_r - kappa_u cos_rj u-hat_r) / n_u          (cos values come from the saved forward result)
__global__ __launch_bounds__(64) void cos_bwd_k1(const float* E, const float* C, const float* cosv, const float* gcos,
                                                 const float* S, const float* CS, int n, int N, int j0, int M, int D,
                                                 float eps_cos, float eps, float* dE, float* DU, float* RNE) {
    const int NM = n * M;                               // local rows per batch
    const int b = blockIdx.x / NM, r = blockIdx.x % NM, jl = r / M, j = j0 + jl, lane = threadIdx.x;
    const float* e = E + ((size_t)b * NM + r) * D;
    const float* s = S + ((size_t)b * n + jl) * D;
    const float* g = gcos + ((size_t)b * NM + r) * N;
    const float* cv = cosv + ((size_t)b * NM + r) * N;
    const float inv_m1 = 1.0f / (float)(M - 1);
    float ee = 0.f, uu = 0.f;
    for (int d = lane; d < D; d += kWave) {
        const float x = e[d], u = (s[d] - x) * inv_m1;
        ee += x * x;
        uu += u * u;
    }
    ee = wave_sum(ee);
    uu = wave_sum(uu);
    float rne, ke, rnu, ku;
    unit_stats(ee, eps_cos, rne, ke);
    unit_stats(uu, eps_cos, rnu, ku);
    float t = 0.f;                                   // g_e . e-hat = sum_k g[r][k] (cos[r][k] - eps)
    for (int k = lane; k < N; k += kWave) t += g[k] * (cv[k] - eps);
    t = wave_sum(t);
    const float gj = g[j], cj = cv[j] - eps;
    if (lane == 0) RNE[(size_t)b * NM + r] = rne;
    for (int d = lane; d < D; d += kWave) {
        const float eh = e[d] * rne, uh = (s[d] - e[d]) * inv_m1 * rnu;
        float ge = gj * uh;
        for (int k = 0; k < N; ++k)
            if (k != j) ge += g[k] * C[((size_t)b * N + k) * D + d] * CS[((size_t)b * N + k) * 2];
        dE[((size_t)b * NM + r) * D + d] = (ge - ke * t * eh) * rne;
        DU[((size_t)b * NM + r) * D + d] = gj * (eh - ku * cj * uh) * rnu;
    }
}

// K2a: dE_r += (sum_i' du_ji' - du_r) / (M - 1)
__global__ __launch_bounds__(256) void cos_bwd_k2a(const float* DU, int speakers, int M, int D, float* dE) {
    const size_t total = (size_t)speakers * D;
    const float inv = 1.0f / (float)(M - 1);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t j = idx / D, d = idx % D;
        const float* du = DU + j * M * D + d;
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += du[(size_t)i * D];
        float* o = dE + j * M * D + d;
        for (int i = 0; i < M; ++i) o[(size_t)i * D] += (s - du[(size_t)i * D]) * inv;
    }
}

// K2b: one wave per centroid k: dC_k = (g_c - kappa_c (g_c . c-hat_k) c-hat_k) / n_c,  g_c = sum_{r not of k} g[r][k] e-hat_r
__global__ __launch_bounds__(64) void cos_bwd_k2b(const float* E, const float* C, const float* cosv, const float* gcos,
                                                  const float* CS, const float* RNE, int n, int N, int j0, int M, int D,
                                                  float eps, float* dC) {
    const int NM = n * M;                               // local rows per batch
    const int b = blockIdx.x / N, k = blockIdx.x % N, lane = threadIdx.x;
    const float rnc = CS[((size_t)b * N + k) * 2], kc = CS[((size_t)b * N + k) * 2 + 1];
    float t = 0.f;
    for (int r = lane; r < NM; r += kWave)
        if (j0 + r / M != k) t += gcos[((size_t)b * NM + r) * N + k] * (cosv[((size_t)b * NM + r) * N + k] - eps);
    t = wave_sum(t);
    for (int d = lane; d < D; d += kWave) {
        float gc = 0.f;
        for (int r = 0; r < NM; ++r)
            if (j0 + r / M != k) gc += gcos[((size_t)b * NM + r) * N + k] * E[((size_t)b * NM + r) * D + d] * RNE[(size_t)b * NM + r];
        const float ch = C[((size_t)b * N + k) * D + d] * rnc;
        dC[((size_t)b * N + k) * D + d] = (gc - kc * t * ch) * rnc;
    }
}

// ---- calc_loss backward: dS[r][k] = gl[b] * dL_r/dS_rk + gp[r] * (the same), one wave per row ----------------------
__global__ __launch_bounds__(256) void calc_loss_bwd_kernel(const float* sim, int B, int n, int N, int j0, int M, float eps,
                                                            float log_eps, int variant, const float* gloss, const float* gper,
                                                            float* dS) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const size_t rows = (size_t)B * n * M;
    for (size_t r = (size_t)blockIdx.x * NW + wid; r < rows; r += (size_t)gridDim.x * NW) {
        const int j = j0 + (int)((r % ((size_t)n * M)) / M);
        const size_t bi = r / ((size_t)n * M);
        const float* row = sim + r * N;
        float* out = dS + r * N;
        const float gr = (gloss ? gloss[bi] : 0.f) + (gper ? gper[r] : 0.f);
        if (variant == 0) {
            float mx = -INFINITY;
            for (int k = lane; k < N; k += kWave) mx = fmaxf(mx, row[k]);
            mx = fmaxf(wave_max(mx), log_eps);
            float z = 0.f;
            for (int k = lane; k < N; k += kWave) z += expf(row[k] - mx);
            z = wave_sum(z) + expf(log_eps - mx);
            const float rz = 1.0f / z;
            for (int k = lane; k < N; k += kWave) out[k] = gr * (expf(row[k] - mx) * rz - (k == j ? 1.0f : 0.0f));
        } else {
            float best = -INFINITY; int besti = 0x7fffffff;
            for (int k = lane; k < N; k += kWave) if (k != j) argmax_merge(best, besti, row[k], k);
            wave_argmax(best, besti);
            const float pos = 1.0f / (1.0f + expf(-row[j]));
            const float neg = N > 1 ? 1.0f / (1.0f + expf(-best)) : 0.f;
            for (int k = lane; k < N; k += kWave)
                out[k] = gr * (k == j ? -pos * (1.0f - pos) : (k == besti ? neg * (1.0f - neg) : 0.0f));
        }
    }
}

int grid_for(size_t items, int per_block) {
    size_t g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

// autograd's side of GE2ELoss.backward: the fused launch has produced dE, dw, db per batch; the incoming gradient g
// ([1] for a single (N,M,D) batch, [B] for a batched loss vector) scales them.  One launch instead of five torch ops:
//   gE[b] = g[b] dE[b];  gw = sum_b g[b] dw[b];  gb = sum_b g[b] db[b]   (the sums in one wave of block 0, fixed order)
__global__ __launch_bounds__(256) void scale_grads_kernel(const float* __restrict__ dE, const float* __restrict__ dw,
                                                          const float* __restrict__ db, const float* __restrict__ g,
                                                          int gB, int B, size_t per_batch, float* __restrict__ gE,
                                                          float* __restrict__ gw, float* __restrict__ gb) {
    if (gE) {
        const size_t total = (size_t)B * per_batch;
        if ((per_batch & 3) == 0) {
            const size_t n4 = total / 4, p4 = per_batch / 4;
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
                const float s = g[gB == 1 ? 0 : i / p4];
                const float4 v = reinterpret_cast<const float4*>(dE)[i];
                reinterpret_cast<float4*>(gE)[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
            }
        } else {
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
                gE[i] = dE[i] * g[gB == 1 ? 0 : i / per_batch];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < kWave && (gw || gb)) {
        float a = 0.f, c = 0.f;
        for (int i = threadIdx.x; i < B; i += kWave) {
            const float s = g[gB == 1 ? 0 : i];
            if (gw) a = fmaf(dw[i], s, a);
            if (gb) c = fmaf(db[i], s, c);
        }
        a = wave_sum(a);
        c = wave_sum(c);
        if (threadIdx.x == 0) {
            if (gw) *gw = a;
            if (gb) *gb = c;
        }
    }
}

}  // namespace

hipError_t launch_scale_grads(const float* dE, const float* dw, const float* db, const float* g, int gB, int B,
                              size_t per_batch, float* gE, float* gw, float* gb, hipStream_t stream) {
    size_t blocks = gE ? ((size_t)B * per_batch / 4 + 255) / 256 : 1;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_grads_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dE, dw, db, g, gB, B, per_batch, gE, gw, gb);
    return hipGetLastError();
}

hipError_t launch_utt_centroids(const float* E, int B, int N, int M, int D, float* U, hipStream_t stream) {
    hipLaunchKernelGGL(utt_centroids_kernel, dim3(grid_for((size_t)B * N * D, 256)), dim3(256), 0, stream, E, B * N, M, D, U);
    return hipGetLastError();
}

hipError_t launch_centroids_bwd(const float* g, int B, int N, int M, int D, float* dE, hipStream_t stream) {
    hipLaunchKernelGGL(centroids_bwd_kernel, dim3(grid_for((size_t)B * N * M * D, 256)), dim3(256), 0, stream, g, B * N, M, D, dE);
    return hipGetLastError();
}

size_t cos_bwd_workspace_bytes(int B, int n, int N, int M, int D) {
    return ((size_t)B * n * D + (size_t)B * n * M * D + (size_t)B * n * M + (size_t)B * N * 2) * sizeof(float);
}

hipError_t launch_cos_bwd(const float* E, const float* C, const float* cosv, const float* gcos, int B, int n, int N, int j0,
                          int M, int D, float eps_cos, float eps, float* dE, float* dC, float* ws, hipStream_t stream) {
    float* S = ws;
    float* DU = S + (size_t)B * n * D;
    float* RNE = DU + (size_t)B * n * M * D;
    float* CS = RNE + (size_t)B * n * M;
    hipLaunchKernelGGL(cos_bwd_k0, dim3(B * (n > N ? n : N)), dim3(64), 0, stream, E, C, n, N, M, D, eps_cos, S, CS);
    hipLaunchKernelGGL(cos_bwd_k1, dim3(B * n * M), dim3(64), 0, stream, E, C, cosv, gcos, S, CS, n, N, j0, M, D, eps_cos, eps,
                       dE, DU, RNE);
    hipLaunchKernelGGL(cos_bwd_k2a, dim3(grid_for((size_t)B * n * D, 256)), dim3(256), 0, stream, DU, B * n, M, D, dE);
    hipLaunchKernelGGL(cos_bwd_k2b, dim3(B * N), dim3(64), 0, stream, E, C, cosv, gcos, CS, RNE, n, N, j0, M, D, eps, dC);
    return hipGetLastError();
}

hipError_t launch_calc_loss_bwd(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant,
                                const float* gloss, const float* gper, float* dS, hipStream_t stream) {
    const float log_eps = eps > 0.f ? logf(eps) : -INFINITY;
    hipLaunchKernelGGL(calc_loss_bwd_kernel, dim3(grid_for((size_t)B * n * M, 4)), dim3(256), 0, stream, sim, B, n, N, j0, M,
                       eps, log_eps, variant, gloss, gper, dS);
    return hipGetLastError();
}

}  // namespace ge2e
