// GE2E_IMPL_WAVE: one WAVE per batch, the whole batch in registers -- the reference's own shapes (training N=2, M=16;
// BASELINE config 1: N=4, M=5; strings/constants.py:98-101): a few dozen rows, where a workgroup per batch spends its
// time in barriers and pipeline ramps (FUSED_SPLIT: 33 k cycles for a 20-row batch).  Exact fp32, no MFMA, no LDS, no
// barriers, no workspace: a batch of 20 x 256 floats is 20 KB in and 20 KB out, so the bound is HBM and the work per
// batch is ~5 k VALU instructions of one wave.
//
// Lane l holds columns 4 l .. 4 l + 3 of every row (D <= 256, D % 4 == 0); a row is one float4 register quadruple.
// Everything that the gradient needs beyond the rows is N float4 per lane (speaker sums, unit centroids, centroid-
// gradient accumulators, leave-one-out accumulators) and wave-uniform scalars.  Dot products are wave reductions
// (DPP + permlane swaps, ge2e_common.hpp); the backward needs none beyond the forward's: (g . x-hat) of every unit-
// vector backward is a combination of cosines already computed.
//
// Per row r = (j, i), with s_j the speaker sum, u_r = (s_j - e_r) / (M - 1), A = w dL/dS:
//   g_e = sum_{k != j} A_rk c-hat_k + A_rj u-hat_r          d e-hat_r = (g_e - kap_e (g_e . e-hat_r) e-hat_r) / n_e
//   gC_k += A_rk e-hat_r (k != j)                            du_r = A_rj (e-hat_r - kap_u cos_rj u-hat_r) / n_u
//   dE_r = d e-hat_r - du_r / (M - 1)  +  [ dc_j / M + (sum_i du_ji) / (M - 1) ]          (the bracket once per speaker)
// M is a template parameter and the speaker loop is unrolled to NX >= N, so every register index is a compile-time
// constant; shapes outside the instantiated (M, NX) table go to the workgroup-per-batch kernels.
#include "ge2e_common.hpp"
#include "ge2e_wave.hpp"

namespace ge2e {

namespace {

__device__ __forceinline__ float dot4w(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
__device__ __forceinline__ float4 mul4(const float4& a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ void fma4(float4& acc, const float4& a, float s) {
    acc.x = fmaf(a.x, s, acc.x); acc.y = fmaf(a.y, s, acc.y); acc.z = fmaf(a.z, s, acc.z); acc.w = fmaf(a.w, s, acc.w);
}

// x / max(|x|, eps) bookkeeping (ge2e_common.hpp: unit_stats) on v_rsq_f32 + one Newton step instead of a square root
// and two IEEE divisions: rn = 1 / max(|x|, eps), kappa = clamped / true norm (0 for a zero vector)
__device__ __forceinline__ void unit_stats_q(float sq, float eps_cos, float eps_cos2, float& rn, float& kappa) {
    const float sqc = fmaxf(sq, eps_cos2);
    float r = __builtin_amdgcn_rsqf(sqc);
    r = r * (1.5f - 0.5f * sqc * r * r);
    rn = r;
    kappa = sq >= eps_cos2 ? 1.0f : (sq > 1e-36f ? eps_cos * __builtin_amdgcn_rsqf(sq) : 0.0f);
}
__device__ __forceinline__ float rcp_q(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}

template <int M, int NX>
__global__ __launch_bounds__(256, 2) void ge2e_wave_kernel(Problem p) {
    const int N = p.N, D = p.D, NM = N * M;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const bool act = 4 * lane < D;
    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const bool contrast = p.variant == 1, want_grad = p.dE != nullptr;
    const float inv_m = 1.0f / (float)M, inv_m1 = 1.0f / (float)(M - 1);
    const float eps_cos2 = eps_cos * eps_cos;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int bi = blockIdx.x * 4 + wid; bi < p.B; bi += gridDim.x * 4) {
        const float* Eb = p.E + (size_t)bi * NM * D + 4 * lane;
        float4 e[NX * M];
#pragma unroll
        for (int j = 0; j < NX; ++j)
#pragma unroll
            for (int i = 0; i < M; ++i)
                e[j * M + i] = (act && j < N) ? *reinterpret_cast<const float4*>(Eb + (size_t)(j * M + i) * D) : z4;

        // speaker sums, unit centroids (s3:34-38 + the cosine's normalisation)
        float4 s[NX], ch[NX];
        float rnc[NX], kc[NX];
        {
            float cc[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                float4 a = e[j * M];
#pragma unroll
                for (int i = 1; i < M; ++i) { a.x += e[j * M + i].x; a.y += e[j * M + i].y; a.z += e[j * M + i].z; a.w += e[j * M + i].w; }
                s[j] = a;
                ch[j] = mul4(a, inv_m);
                cc[j] = dot4w(ch[j], ch[j]);
            }
            wave_sum_n<NX>(cc);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                unit_stats_q(cc[j], eps_cos, eps_cos2, rnc[j], kc[j]);
                ch[j] = mul4(ch[j], rnc[j]);
            }
        }

        float4 gC[NX], DU[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) { gC[j] = z4; DU[j] = z4; }
        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;

#pragma unroll
        for (int j = 0; j < NX; ++j) {
            if (j < N) {
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const int r = j * M + i;
                    const float4 x = e[r];
                    // leave-one-out centroid of the own speaker (s3:96-112)
                    const float4 u = make_float4((s[j].x - x.x) * inv_m1, (s[j].y - x.y) * inv_m1, (s[j].z - x.z) * inv_m1,
                                                 (s[j].w - x.w) * inv_m1);
                    // every dot product of the row on the RAW vectors, reduced together; the norms scale them afterwards
                    float dt[NX + 2];   // x . c-hat_k (k != j; slot j: x . u), x . x, u . u
#pragma unroll
                    for (int k = 0; k < NX; ++k) dt[k] = k == j ? dot4w(x, u) : dot4w(x, ch[k]);
                    dt[NX] = dot4w(x, x);
                    dt[NX + 1] = dot4w(u, u);
                    wave_sum_n<NX + 2>(dt);
                    float rne, ke, rnu, ku;
                    unit_stats_q(dt[NX], eps_cos, eps_cos2, rne, ke);
                    unit_stats_q(dt[NX + 1], eps_cos, eps_cos2, rnu, ku);
                    const float4 eh = mul4(x, rne);
                    const float4 uh = mul4(u, rnu);
                    const float cosd = dt[j] * rne * rnu;
                    float c0[NX];
#pragma unroll
                    for (int k = 0; k < NX; ++k) c0[k] = k == j ? cosd : (k < N ? dt[k] * rne : 0.f);

                    // eq. (6) / eq. (7) on the row's N similarities (wave-uniform scalars)
                    const float sjj = fmaf(w, cosd + eps, bias);
                    float g[NX], per;
                    if (!contrast) {
                        float sv[NX], mx = log_eps;
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            sv[k] = k < N ? fmaf(w, c0[k] + eps, bias) : -INFINITY;
                            mx = fmaxf(mx, sv[k]);
                        }
                        float zoff = __expf(log_eps - mx);   // the "+ small_err" inside the log (s3:120)
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            g[k] = __expf(sv[k] - mx);        // exp(-inf) = 0 for k >= N
                            if (k != j) zoff += g[k];
                        }
                        const float z = zoff + __expf(sjj - mx);
                        per = (mx - sjj) + __logf(z);
                        const float rz = rcp_q(z);
#pragma unroll
                        for (int k = 0; k < NX; ++k) g[k] = k == j ? -zoff * rz : g[k] * rz;
                    } else {
                        float best = -INFINITY;
                        int besti = -1;
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            const float sk = fmaf(w, c0[k] + eps, bias);
                            if (k != j && k < N && sk > best) { best = sk; besti = k; }
                        }
                        const float pos = rcp_q(1.0f + __expf(-sjj));
                        const float neg = (N > 1) ? rcp_q(1.0f + __expf(-best)) : 0.0f;
                        per = 1.0f - pos + neg;
#pragma unroll
                        for (int k = 0; k < NX; ++k) g[k] = k == j ? -pos * (1.0f - pos) : (k == besti ? neg * (1.0f - neg) : 0.f);
                    }
                    loss_acc += per;
                    if (p.per && lane == 0) p.per[(size_t)bi * NM + r] = per;
                    float coef = 0.f;
#pragma unroll
                    for (int k = 0; k < NX; ++k) {
                        if (k < N) {
                            dw_acc = fmaf(g[k], c0[k] + eps, dw_acc);
                            db_acc += g[k];
                            coef = fmaf(w * g[k], c0[k], coef);   // g_e . e-hat_r: every dot product is a cosine we have
                        }
                    }
                    if (want_grad) {
                        const float ad = w * g[j];
                        float4 ge = mul4(uh, ad);
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            if (k != j && k < N) {
                                const float a = w * g[k];
                                fma4(ge, ch[k], a);
                                fma4(gC[k], eh, a);
                            }
                        }
                        const float t = ke * coef;
                        const float4 de = make_float4((ge.x - t * eh.x) * rne, (ge.y - t * eh.y) * rne, (ge.z - t * eh.z) * rne,
                                                      (ge.w - t * eh.w) * rne);
                        const float tu = ku * cosd;          // (g_u . u-hat) = A_rj cos_rj
                        const float sc = ad * rnu;
                        const float4 du = make_float4((eh.x - tu * uh.x) * sc, (eh.y - tu * uh.y) * sc, (eh.z - tu * uh.z) * sc,
                                                      (eh.w - tu * uh.w) * sc);
                        DU[j].x += du.x; DU[j].y += du.y; DU[j].z += du.z; DU[j].w += du.w;
                        e[r] = make_float4(de.x - du.x * inv_m1, de.y - du.y * inv_m1, de.z - du.z * inv_m1, de.w - du.w * inv_m1);
                    }
                }
            }
        }

        if (want_grad) {
            float* Gb = p.dE + (size_t)bi * NM * D + 4 * lane;
            float gd[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) gd[j] = dot4w(gC[j], ch[j]);
            wave_sum_n<NX>(gd);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if (j < N) {
                    // centroid gradient through its normalisation, + the leave-one-out sums: one row per speaker
                    const float t = kc[j] * gd[j];
                    const float a = rnc[j] * inv_m;
                    const float4 kj = make_float4(fmaf(gC[j].x - t * ch[j].x, a, DU[j].x * inv_m1), fmaf(gC[j].y - t * ch[j].y, a, DU[j].y * inv_m1),
                                                  fmaf(gC[j].z - t * ch[j].z, a, DU[j].z * inv_m1), fmaf(gC[j].w - t * ch[j].w, a, DU[j].w * inv_m1));
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const int r = j * M + i;
                        if (act)
                            *reinterpret_cast<float4*>(Gb + (size_t)r * D) =
                                make_float4(e[r].x + kj.x, e[r].y + kj.y, e[r].z + kj.z, e[r].w + kj.w);
                    }
                }
            }
        }
        if (lane == 0) {
            p.loss[bi] = loss_acc;
            if (p.dw) p.dw[bi] = dw_acc;
            if (p.db) p.db[bi] = db_acc;
        }
    }
}

struct WaveShape { int M, NX; };
constexpr WaveShape kShapes[] = {{2, 6}, {3, 5}, {4, 4}, {5, 4}, {6, 3}, {8, 3}, {10, 2}, {16, 2}};

template <int M, int NX>
hipError_t launch_mn(const Problem& p, hipStream_t stream) {
    const int blocks_needed = (p.B + 3) / 4;
    int grid = device_cu_count() * 2;   // two workgroups (8 waves) per CU at <= 256 VGPRs
    if (grid > blocks_needed) grid = blocks_needed;
    hipLaunchKernelGGL((ge2e_wave_kernel<M, NX>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace

bool wave_supports(int N, int M, int D) {
    if (D < 4 || D > 256 || (D & 3) || N < 1 || M < 2) return false;
    for (const WaveShape& s : kShapes)
        if (s.M == M) return N <= s.NX;
    return false;
}

hipError_t launch_wave(const Problem& p, hipStream_t stream) {
    switch (p.M) {
        case 2: return launch_mn<2, 6>(p, stream);
        case 3: return launch_mn<3, 5>(p, stream);
        case 4: return launch_mn<4, 4>(p, stream);
        case 5: return launch_mn<5, 4>(p, stream);
        case 6: return launch_mn<6, 3>(p, stream);
        case 8: return launch_mn<8, 3>(p, stream);
        case 10: return launch_mn<10, 2>(p, stream);
        case 16: return launch_mn<16, 2>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ge2e
