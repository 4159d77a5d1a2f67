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

// Three passes per batch.  (1) per row: its N + 2 dot products on the raw vectors, reduced to scalars and dropped into
// LANE r of N + 2 registers.  (2) ONCE for all rows, lane r = row r: norms, cosines, softmax / contrast, dL/dS, the
// coefficients of the row's gradient (the part that every lane used to compute redundantly per row: ~60 of a row's
// ~350 instructions).  (3) per row: its coefficients read back as scalars (v_readlane), the vector part of the gradient.
// RAW (ge2e_loss_fwd_bwd_raw): the rows come from the encoder's raw projection Y through the gather index
// src and are L2-normalised in the load stage (s2:34, s4:186-189); the store stage pushes dL/dE through that normalisation,
// dY_r = (g - e (e . g)) / |y|, and scatters it back.  Two more wave reductions a row (|y|^2 on the way in, e . g on the way
// out) and a second read of the row, which is an L2 hit (the batch was read microseconds earlier).
template <int M, int NX, bool RAW = false>
__global__ __launch_bounds__(256, (4 * NX * M + 16 * NX > 200 ? 1 : 2)) void ge2e_wave_kernel(Problem p) {   // large: one wave per
    static_assert(NX * M <= 64, "a row per lane in pass 2");                                                  // SIMD, AGPRs instead of scratch
    const int N_outer = p.N, D_outer = p.D;
    const int lane_outer = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const bool contrast = p.variant == 1, want_grad = p.dE != nullptr;
    const float inv_m = 1.0f / (float)M, inv_m1 = 1.0f / (float)(M - 1);
    const float eps_cos2 = eps_cos * eps_cos;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int bi = blockIdx.x * 4 + wid; bi < p.B; bi += gridDim.x * 4) {
        // shape- and lane-derived values re-derived per batch from opaque copies: as loop invariants hipcc keeps every
        // row offset and predicate mask in SGPRs, runs out, and parks them in VGPR lanes (ge2e_team.hip, hazard 8)
        int N = N_outer, D = D_outer, lane = lane_outer;
        asm volatile("" : "+s"(N), "+s"(D), "+v"(lane));
        const int NM = N * M;
        const bool act = 4 * lane < D;
        const int jl = lane / M;                 // the speaker of "my" row in pass 2
        const bool rowv = lane < NM;
        const float* Eb = p.E + (size_t)bi * NM * D + 4 * lane;
        float4 e[NX * M];
        int srow = 0;         // RAW: lane r = the row of Y that is row r of the (N,M,D) block
        int sok = 1;          // RAW: ... and whether src named a row inside the batch (a caller's unvalidated index tensor:
                              // an entry outside range(N M) reads a clamped row and its dY store is dropped)
        float rny = 0.f;      // RAW: lane r = 1 / |y_r|
        if (!RAW) {
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i = 0; i < M; ++i)
                    e[j * M + i] = (act && j < N) ? *reinterpret_cast<const float4*>(Eb + (size_t)(j * M + i) * D) : z4;
        } else {
            srow = rowv ? (p.src ? p.src[(size_t)bi * NM + lane] : lane) : 0;
            sok = (unsigned)srow < (unsigned)NM;
            srow = min(max(srow, 0), NM - 1);
#pragma unroll
            for (int j = 0; j < NX; ++j)
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const int sr = __builtin_amdgcn_readlane(srow, j * M + i);
                    e[j * M + i] = (act && j < N) ? *reinterpret_cast<const float4*>(Eb + (size_t)sr * D) : z4;
                }
            // e = y / |y| (s2:34: a plain division, no epsilon), the speaker's M norms reduced together
            static_for<0, NX>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < N) {
                    float nn[M];
#pragma unroll
                    for (int i = 0; i < M; ++i) nn[i] = dot4w(e[j * M + i], e[j * M + i]);
                    wave_sum_to_sgpr<M>(nn);
                    static_for<0, M>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        float r = __builtin_amdgcn_rsqf(nn[i]);
                        r = r * (1.5f - 0.5f * nn[i] * r * r);
                        e[j * M + i] = mul4(e[j * M + i], r);
                        rny = lane_put<j * M + i>(rny, r);
                    });
                }
            });
        }

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
            wave_sum_to_sgpr<NX>(cc);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                unit_stats_q(cc[j], eps_cos, eps_cos2, rnc[j], kc[j]);
                ch[j] = mul4(ch[j], rnc[j]);
            }
        }

        // ---- pass 1: T[k] lane r = x_r . c-hat_k (k != j_r) | x_r . u_r (k = j_r);  T[NX] = |x_r|^2;  T[NX + 1] = |u_r|^2
        float T[NX + 2];
#pragma unroll
        for (int k = 0; k < NX + 2; ++k) T[k] = 0.f;
        static_for<0, NX * M>([&](auto rc) {
            constexpr int r = decltype(rc)::value, j = r / M;
            if (j < N) {
                const float4 x = e[r];
                // leave-one-out centroid of the own speaker (s3:96-112)
                const float4 u = make_float4((s[j].x - x.x) * inv_m1, (s[j].y - x.y) * inv_m1, (s[j].z - x.z) * inv_m1,
                                             (s[j].w - x.w) * inv_m1);
                float dt[NX + 2];
#pragma unroll
                for (int k = 0; k < NX; ++k) dt[k] = k == j ? dot4w(x, u) : dot4w(x, ch[k]);
                dt[NX] = dot4w(x, x);
                dt[NX + 1] = dot4w(u, u);
                wave_sum_to_sgpr<NX + 2>(dt);
#pragma unroll
                for (int k = 0; k < NX + 2; ++k) T[k] = lane_put<r>(T[k], dt[k]);
            }
            if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two rows at a time: more in flight costs registers
        });

        // ---- pass 2: lane r = row r
        float rne, ke, rnu, ku;
        unit_stats_q(T[NX], eps_cos, eps_cos2, rne, ke);
        unit_stats_q(T[NX + 1], eps_cos, eps_cos2, rnu, ku);
        float c0[NX], g[NX];
        float cosd = 0.f;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const bool own = k == jl;
            c0[k] = k < N ? T[k] * rne * (own ? rnu : 1.0f) : 0.f;
            cosd = own ? c0[k] : cosd;
        }
        const float sjj = fmaf(w, cosd + eps, bias);
        float per;
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
                zoff += k == jl ? 0.f : g[k];
            }
            const float z = zoff + __expf(sjj - mx);
            per = (mx - sjj) + __logf(z);
            const float rz = rcp_q(z);
#pragma unroll
            for (int k = 0; k < NX; ++k) g[k] = k == jl ? -zoff * rz : g[k] * rz;
        } else {
            float best = -INFINITY;
            int besti = -1;
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const float sk = fmaf(w, c0[k] + eps, bias);
                if (k != jl && k < N && sk > best) { best = sk; besti = k; }
            }
            const float pos = rcp_q(1.0f + __expf(-sjj));
            const float neg = (N > 1) ? rcp_q(1.0f + __expf(-best)) : 0.0f;
            per = 1.0f - pos + neg;
#pragma unroll
            for (int k = 0; k < NX; ++k) g[k] = k == jl ? -pos * (1.0f - pos) : (k == besti ? neg * (1.0f - neg) : 0.f);
        }
        float red[3] = {0.f, 0.f, 0.f};   // loss, dw, db of my row
        float coef = 0.f;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (k < N) {
                red[1] = fmaf(g[k], c0[k] + eps, red[1]);
                red[2] += g[k];
                coef = fmaf(w * g[k], c0[k], coef);   // g_e . e-hat_r: every dot product is a cosine we have
            }
            g[k] *= w;                                // A = w dL/dS
        }
        red[0] = per;
        if (!rowv) { red[0] = 0.f; red[1] = 0.f; red[2] = 0.f; }
        if (p.per && rowv) p.per[(size_t)bi * NM + lane] = per;
        float adl = 0.f;
#pragma unroll
        for (int k = 0; k < NX; ++k) adl = k == jl ? g[k] : adl;
        const float tl = ke * coef;                   // kap_e (g_e . e-hat)
        const float tul = ku * cosd;                  // kap_u cos_rj: (g_u . u-hat) / A_rj
        const float scl = adl * rnu;
        wave_sum_to_sgpr<3>(red);

        // ---- pass 3: the vector part of every row's gradient
        float4 gC[NX], DU[NX];
#pragma unroll
        for (int j = 0; j < NX; ++j) { gC[j] = z4; DU[j] = z4; }
        if (want_grad) {
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if (j < N) {
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        const int r = j * M + i;
                        const float4 x = e[r];
                        const float rne_r = lane_get(rne, r), rnu_r = lane_get(rnu, r);
                        const float4 eh = mul4(x, rne_r);
                        const float4 uh = make_float4((s[j].x - x.x) * (inv_m1 * rnu_r), (s[j].y - x.y) * (inv_m1 * rnu_r),
                                                      (s[j].z - x.z) * (inv_m1 * rnu_r), (s[j].w - x.w) * (inv_m1 * rnu_r));
                        float4 ge = mul4(uh, lane_get(adl, r));
#pragma unroll
                        for (int k = 0; k < NX; ++k) {
                            if (k != j && k < N) {
                                const float a = lane_get(g[k], r);
                                fma4(ge, ch[k], a);
                                fma4(gC[k], eh, a);
                            }
                        }
                        const float t = lane_get(tl, r), tu = lane_get(tul, r), sc = lane_get(scl, r);
                        const float4 de = make_float4((ge.x - t * eh.x) * rne_r, (ge.y - t * eh.y) * rne_r, (ge.z - t * eh.z) * rne_r,
                                                      (ge.w - t * eh.w) * rne_r);
                        const float4 du = make_float4((eh.x - tu * uh.x) * sc, (eh.y - tu * uh.y) * sc, (eh.z - tu * uh.z) * sc,
                                                      (eh.w - tu * uh.w) * sc);
                        DU[j].x += du.x; DU[j].y += du.y; DU[j].z += du.z; DU[j].w += du.w;
                        e[r] = make_float4(de.x - du.x * inv_m1, de.y - du.y * inv_m1, de.z - du.z * inv_m1, de.w - du.w * inv_m1);
                        if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            float* Gb = p.dE + (size_t)bi * NM * D + 4 * lane;
            float gd[NX];
#pragma unroll
            for (int j = 0; j < NX; ++j) gd[j] = dot4w(gC[j], ch[j]);
            wave_sum_to_sgpr<NX>(gd);
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if (j < N) {
                    // centroid gradient through its normalisation, + the leave-one-out sums: one row per speaker
                    const float t = kc[j] * gd[j];
                    const float a = rnc[j] * inv_m;
                    const float4 kj = make_float4(fmaf(gC[j].x - t * ch[j].x, a, DU[j].x * inv_m1), fmaf(gC[j].y - t * ch[j].y, a, DU[j].y * inv_m1),
                                                  fmaf(gC[j].z - t * ch[j].z, a, DU[j].z * inv_m1), fmaf(gC[j].w - t * ch[j].w, a, DU[j].w * inv_m1));
                    if (!RAW) {
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            const int r = j * M + i;
                            if (act)
                                *reinterpret_cast<float4*>(Gb + (size_t)r * D) =
                                    make_float4(e[r].x + kj.x, e[r].y + kj.y, e[r].z + kj.z, e[r].w + kj.w);
                        }
                    } else {
                        // dY_r = (g - e (e . g)) / |y| with g = dE_r, scattered to the row it came from; e is rebuilt from a
                        // second read of y (the registers that held it carry the gradient now)
                        float4 yh[M];
                        float eg[M];
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            const int r = j * M + i;
                            const int sr = __builtin_amdgcn_readlane(srow, r);
                            const float4 y = act ? *reinterpret_cast<const float4*>(Eb + (size_t)sr * D) : z4;
                            yh[i] = mul4(y, lane_get(rny, r));
                            e[r] = make_float4(e[r].x + kj.x, e[r].y + kj.y, e[r].z + kj.z, e[r].w + kj.w);
                            eg[i] = dot4w(yh[i], e[r]);
                        }
                        wave_sum_to_sgpr<M>(eg);
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            const int r = j * M + i;
                            const int sr = __builtin_amdgcn_readlane(srow, r);
                            const float rn = lane_get(rny, r), t = eg[i];
                            if (act && __builtin_amdgcn_readlane(sok, r))
                                *reinterpret_cast<float4*>(Gb + (size_t)sr * D) =
                                    make_float4((e[r].x - t * yh[i].x) * rn, (e[r].y - t * yh[i].y) * rn,
                                                (e[r].z - t * yh[i].z) * rn, (e[r].w - t * yh[i].w) * rn);
                        }
                    }
                }
            }
        }
        if (lane == 0) {
            p.loss[bi] = red[0];
            if (p.dw) p.dw[bi] = red[1];
            if (p.db) p.db[bi] = red[2];
        }
    }
}

// Two instantiations per M: a small one (<= 30 rows: 2 waves per SIMD, registers only) and a large one (<= 64 rows, the
// lane-per-row limit of pass 2: one wave per SIMD, the row registers spill into AGPRs).
struct WaveShape { int M, NX, NXL; };
constexpr WaveShape kShapes[] = {{2, 6, 12}, {3, 5, 10}, {4, 4, 10}, {5, 4, 8}, {6, 3, 8}, {8, 3, 8}, {10, 2, 6}, {16, 2, 3}};

template <int M, int NX, bool RAW>
hipError_t launch_mn(const Problem& p, hipStream_t stream) {
    const int blocks_needed = (p.B + 3) / 4;
    int grid = device_cu_count() * 2;   // two workgroups (8 waves) per CU at <= 256 VGPRs
    if (grid > blocks_needed) grid = blocks_needed;
    hipLaunchKernelGGL((ge2e_wave_kernel<M, NX, RAW>), dim3(grid), dim3(256), 0, stream, p);
    return hipGetLastError();
}
template <int M, int NX, int NXL>
hipError_t launch_m(const Problem& p, hipStream_t stream) {
    if (p.raw) return p.N <= NX ? launch_mn<M, NX, true>(p, stream) : hipErrorInvalidValue;   // raw rows: the small shapes only
    return p.N <= NX ? launch_mn<M, NX, false>(p, stream) : launch_mn<M, NXL, false>(p, stream);
}

}  // namespace

bool wave_supports(int N, int M, int D) {
    if (D < 4 || D > 256 || (D & 3) || N < 1 || M < 2) return false;
    for (const WaveShape& s : kShapes)
        if (s.M == M) return N <= s.NXL;
    return false;
}

// the large instantiations run one wave per SIMD and a single batch takes 30-45 us on its one wave: they pay from a few
// hundred batches per launch on (the workgroup-per-batch kernel needs 27 us per 256 batches)
// ge2e_loss_fwd_bwd_raw: the register-only instantiations (a few dozen rows: the reference's training shapes)
bool wave_supports_raw(int N, int M, int D) { return wave_supports(N, M, D) && !wave_is_large(N, M); }

bool wave_is_large(int N, int M) {
    for (const WaveShape& s : kShapes)
        if (s.M == M) return N > s.NX;
    return false;
}

hipError_t launch_wave(const Problem& p, hipStream_t stream) {
    switch (p.M) {
        case 2: return launch_m<2, 6, 12>(p, stream);
        case 3: return launch_m<3, 5, 10>(p, stream);
        case 4: return launch_m<4, 4, 10>(p, stream);
        case 5: return launch_m<5, 4, 8>(p, stream);
        case 6: return launch_m<6, 3, 8>(p, stream);
        case 8: return launch_m<8, 3, 8>(p, stream);
        case 10: return launch_m<10, 2, 6>(p, stream);
        case 16: return launch_m<16, 2, 3>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ge2e
