// GE2E_IMPL_GENERIC: one 256-thread workgroup per (N,M,D) batch, exact fp32 on the
// VALU, any shape.  It is the correctness anchor on the GPU (simple enough to audit
// against oracle/ge2e_oracle.py:closed_form line by line) and the fallback for shapes
// the fused kernels do not cover.  Intermediates live in a per-workgroup slice of the
// caller's workspace (L2-resident); the phases are separated by __syncthreads().
//
// Reference semantics restated here (embedding_model_GE2E/s3_loss_function_GE2E.py):
//   phase A  s3:34-38   centroids (mean over M), normalised once
//   phase B  s3:42-80   cos matrix: leave-one-out centroid on the own-speaker column
//                        (s3:96-112), +small_err everywhere (s3:79);
//            s3:27      S = w*cos + b;   s3:115-127  per-row log(sum exp S + eps) - S_jj
//   phase C/D s4:200    the gradient autograd would produce (dE, dw, db)
#include "ge2e_common.hpp"
#include "ge2e_generic.hpp"

namespace ge2e {

namespace {
constexpr int RS_RNE = 0, RS_KE = 1, RS_RNU = 2, RS_KU = 3, RS_COSD = 4, RS_AD = 5, RS_COEF = 6;
constexpr int kMaxWaves = 16;
}  // namespace

__global__ __launch_bounds__(256) void ge2e_generic_kernel(Problem p, size_t ws_stride) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int NW = blockDim.x >> 6;
    const int N = p.N, M = p.M, D = p.D, NM = N * M;
    const GenericLayout L = generic_layout(N, M, D);
    float* ws = p.ws + (size_t)blockIdx.x * ws_stride;
    float* CH = ws + L.ch;    // [N][D]    unit centroids
    float* CHT = ws + L.cht;  // [D][npad] same, transposed (lane <-> centroid reads)
    float* SS = ws + L.ss;    // [N][D]    per-speaker sums
    float* GC = ws + L.gc;    // [N][D]    dL/d c-hat, then dL/dc
    float* A = ws + L.a;      // [NM][N]   cos, then dL/dcos with the diagonal zeroed
    float* RST = ws + L.rowstat;
    float* CST = ws + L.cstat;
    const int npad = L.npad;
    __shared__ float red[3][kMaxWaves];

    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const float fM = (float)M, fM1 = (float)(M - 1);
    const bool contrast = p.variant == 1;

    for (int bi = blockIdx.x; bi < p.B; bi += gridDim.x) {
        const float* E = p.E + (size_t)bi * NM * D;

        // ---- phase A: speaker sums and unit centroids ------------------------------
        for (int j = wid; j < N; j += NW) {
            float sq = 0.f;
            for (int d = lane; d < D; d += kWave) {
                float s = 0.f;
                for (int i = 0; i < M; ++i) s += E[(size_t)(j * M + i) * D + d];
                SS[j * D + d] = s;
                float c = s / fM;
                sq += c * c;
            }
            sq = wave_sum(sq);
            float rn, kap;
            unit_stats(sq, eps_cos, rn, kap);
            for (int d = lane; d < D; d += kWave) {
                float c = SS[j * D + d] / fM * rn;
                CH[j * D + d] = c;
                CHT[(size_t)d * npad + j] = c;
            }
            if (lane == 0) { CST[j * 4 + 0] = rn; CST[j * 4 + 1] = kap; }
        }
        __syncthreads();

        // ---- phase B: cos rows, loss, dL/dcos ----------------------------------------
        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        for (int r = wid; r < NM; r += NW) {
            const int j = r / M;
            const float* er = E + (size_t)r * D;
            float* Arow = A + (size_t)r * N;
            float ee = 0.f, uu = 0.f, eu = 0.f;
            for (int d = lane; d < D; d += kWave) {
                float e = er[d];
                float u = (SS[j * D + d] - e) / fM1;
                ee += e * e; uu += u * u; eu += e * u;
            }
            ee = wave_sum(ee); uu = wave_sum(uu); eu = wave_sum(eu);
            float rne, ke, rnu, ku;
            unit_stats(ee, eps_cos, rne, ke);
            unit_stats(uu, eps_cos, rnu, ku);
            const float cosd = eu * rne * rnu;
            const float sjj = w * (cosd + eps) + bias;

            // pass 1: lane <-> centroid k, serial over d
            float mx = -INFINITY;
            float best = -INFINITY; int besti = 0x7fffffff;
            for (int kc = 0; kc < N; kc += kWave) {
                const int k = kc + lane;
                const int kk = k < N ? k : N - 1;
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc = fmaf(er[d], CHT[(size_t)d * npad + kk], acc);
                const float c0 = (k == j) ? cosd : acc * rne;
                if (k < N) {
                    Arow[k] = c0;
                    if (p.cos_out) p.cos_out[((size_t)bi * NM + r) * N + k] = c0 + eps;
                    const float s = w * (c0 + eps) + bias;
                    mx = fmaxf(mx, s);
                    if (k != j && s > best) { best = s; besti = k; }
                }
            }

            float per, coef = 0.f, ad = 0.f;
            if (!contrast) {
                mx = fmaxf(wave_max(mx), log_eps);
                // z_off = everything except the own-speaker term: 1 - p_jj = z_off / z has no
                // cancellation when the softmax is peaked on the diagonal (trained embeddings)
                float zoff = 0.f;
                for (int k = lane; k < N; k += kWave)
                    if (k != j) zoff += expf(w * (Arow[k] + eps) + bias - mx);
                zoff = wave_sum(zoff) + expf(log_eps - mx);
                const float z = zoff + expf(sjj - mx);
                per = (mx - sjj) + logf(z);
                const float rz = 1.0f / z;
                for (int k = lane; k < N; k += kWave) {
                    const float c0 = Arow[k];
                    const float g = (k == j) ? -zoff * rz : expf(w * (c0 + eps) + bias - mx) * rz;
                    dw_acc += g * (c0 + eps);
                    db_acc += g;
                    const float a = w * g;
                    coef += a * c0;
                    if (k == j) { ad = a; Arow[k] = 0.f; } else { Arow[k] = a; }
                }
            } else {
                wave_argmax(best, besti);
                const float pos = 1.0f / (1.0f + expf(-sjj));
                const float neg = (N > 1) ? 1.0f / (1.0f + expf(-best)) : 0.0f;
                per = 1.0f - pos + neg;
                for (int k = lane; k < N; k += kWave) {
                    const float c0 = Arow[k];
                    float g = 0.f;
                    if (k == j) g = -pos * (1.0f - pos);
                    else if (k == besti) g = neg * (1.0f - neg);
                    dw_acc += g * (c0 + eps);
                    db_acc += g;
                    const float a = w * g;
                    coef += a * c0;
                    if (k == j) { ad = a; Arow[k] = 0.f; } else { Arow[k] = a; }
                }
            }
            coef = wave_sum(coef);
            ad = wave_sum(ad);
            loss_acc += per;
            if (lane == 0) {
                if (p.per) p.per[(size_t)bi * NM + r] = per;
                float* rs = RST + (size_t)r * 8;
                rs[RS_RNE] = rne; rs[RS_KE] = ke; rs[RS_RNU] = rnu; rs[RS_KU] = ku;
                rs[RS_COSD] = cosd; rs[RS_AD] = ad; rs[RS_COEF] = coef;
            }
        }
        dw_acc = wave_sum(dw_acc);
        db_acc = wave_sum(db_acc);
        if (lane == 0) { red[0][wid] = loss_acc; red[1][wid] = dw_acc; red[2][wid] = db_acc; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float l = 0.f, a = 0.f, c = 0.f;
            for (int i = 0; i < NW; ++i) { l += red[0][i]; a += red[1][i]; c += red[2][i]; }
            if (p.loss) p.loss[bi] = l;
            if (p.dw) p.dw[bi] = a;
            if (p.db) p.db[bi] = c;
        }

        if (p.dE) {
            float* dE = p.dE + (size_t)bi * NM * D;
            // ---- phase C: dL/d c-hat = A_off^T . E-hat, then through the centroid norm --
            for (int k = wid; k < N; k += NW) {
                float coef = 0.f;
                for (int d = lane; d < D; d += kWave) {
                    float acc = 0.f;
                    for (int r = 0; r < NM; ++r)
                        acc = fmaf(A[(size_t)r * N + k] * RST[(size_t)r * 8 + RS_RNE], E[(size_t)r * D + d], acc);
                    GC[k * D + d] = acc;
                    coef += acc * CH[k * D + d];
                }
                coef = wave_sum(coef);
                const float rn = CST[k * 4 + 0], kap = CST[k * 4 + 1];
                for (int d = lane; d < D; d += kWave)
                    GC[k * D + d] = (GC[k * D + d] - kap * coef * CH[k * D + d]) * rn;
            }
            __syncthreads();
            // ---- phase D: dE rows, one speaker per wave -----------------------------------
            for (int j = wid; j < N; j += NW) {
                for (int d = lane; d < D; d += kWave) {
                    const float s = SS[j * D + d];
                    const float dcj = GC[j * D + d] / fM;
                    float dusum = 0.f;
                    for (int i = 0; i < M; ++i) {
                        const int r = j * M + i;
                        const float* rs = RST + (size_t)r * 8;
                        const float e = E[(size_t)r * D + d];
                        const float eh = e * rs[RS_RNE];
                        const float uh = (s - e) / fM1 * rs[RS_RNU];
                        dusum += rs[RS_AD] * (eh - rs[RS_KU] * rs[RS_COSD] * uh) * rs[RS_RNU];
                    }
                    for (int i = 0; i < M; ++i) {
                        const int r = j * M + i;
                        const float* rs = RST + (size_t)r * 8;
                        const float e = E[(size_t)r * D + d];
                        const float eh = e * rs[RS_RNE];
                        const float uh = (s - e) / fM1 * rs[RS_RNU];
                        const float du = rs[RS_AD] * (eh - rs[RS_KU] * rs[RS_COSD] * uh) * rs[RS_RNU];
                        float g = rs[RS_AD] * uh;
                        const float* Arow = A + (size_t)r * N;
                        for (int k = 0; k < N; ++k) g = fmaf(Arow[k], CH[k * D + d], g);
                        dE[(size_t)r * D + d] =
                            (g - rs[RS_KE] * rs[RS_COEF] * eh) * rs[RS_RNE] + dcj + (dusum - du) / fM1;
                    }
                }
            }
        }
        __syncthreads();  // workspace slice is reused by the next batch of this workgroup
    }
}

__global__ void ge2e_centroids_kernel(const float* E, int rows /*B*N*/, int M, int D, float* cent) {
    const size_t total = (size_t)rows * D;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const size_t jr = idx / D;
        const int d = (int)(idx - jr * D);
        float s = 0.f;
        for (int i = 0; i < M; ++i) s += E[(jr * M + i) * D + d];
        cent[idx] = s / (float)M;
    }
}

// get_cos_sim with CALLER-SUPPLIED centroids (s3:42-80 uses its `centroids` argument for every other-speaker
// column and the leave-one-out centroid of `embeddings` on the own-speaker column): one wave per row r = (j, i).
//   cos[r][k] = e_r . c_k / (max(|e_r|, eps_cos) max(|c_k|, eps_cos)) + eps            (k != j)
//   cos[r][j] = e_r . u_r / (max(|e_r|, eps_cos) max(|u_r|, eps_cos)) + eps,  u_r = (sum_i' e_ji' - e_r) / (M - 1)
// E [B][n][M][D], C [B][N][D] -> cos [B][n][M][N].  A forward-only helper (the eval script's path), not a hot path.
// LOCAL ROWS (the speaker-sharded loss): E holds the rows of the n speakers j0 .. j0 + n - 1 of a batch of N
// speakers whose centroids are all in C; the own-speaker column of local speaker jl is j0 + jl.  n = N, j0 = 0 is the
// reference's get_cos_sim(embeddings, centroids).
__global__ __launch_bounds__(256) void ge2e_cos_centroids_kernel(const float* E, const float* C, int B, int n, int N, int j0,
                                                                  int M, int D, float eps_cos, float eps, float* cos) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const size_t rows = (size_t)B * n * M;
    for (size_t r = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6); r < rows; r += (size_t)gridDim.x * wpb) {
        const size_t bj = r / M;                    // (batch, local speaker)
        const int j = j0 + (int)(bj % n);           // its column
        const size_t bi = bj / n;
        const float* e = E + r * D;
        const float* spk = E + bj * (size_t)M * D;
        const float inv_m1 = 1.0f / (float)(M - 1);
        float ee = 0.f, eu = 0.f, uu = 0.f;
        for (int d = lane; d < D; d += 64) {
            float s = 0.f;
            for (int i = 0; i < M; ++i) s += spk[(size_t)i * D + d];
            const float x = e[d], u = (s - x) * inv_m1;
            ee = fmaf(x, x, ee); eu = fmaf(x, u, eu); uu = fmaf(u, u, uu);
        }
        ee = wave_sum(ee); eu = wave_sum(eu); uu = wave_sum(uu);
        const float ne = fmaxf(sqrtf(ee), eps_cos);
        for (int k = 0; k < N; ++k) {
            float out;
            if (k == j) {
                out = eu / (ne * fmaxf(sqrtf(uu), eps_cos));
            } else {
                const float* c = C + (bi * N + k) * (size_t)D;
                float ec = 0.f, cc = 0.f;
                for (int d = lane; d < D; d += 64) { const float y = c[d]; ec = fmaf(e[d], y, ec); cc = fmaf(y, y, cc); }
                ec = wave_sum(ec); cc = wave_sum(cc);
                out = ec / (ne * fmaxf(sqrtf(cc), eps_cos));
            }
            if (lane == 0) cos[r * N + k] = out + eps;
        }
    }
}

hipError_t launch_cos_centroids(const float* E, const float* C, int B, int n, int N, int j0, int M, int D, float eps_cos,
                                float eps, float* cos, hipStream_t stream) {
    const size_t rows = (size_t)B * n * M;
    const int grid = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(ge2e_cos_centroids_kernel, dim3(grid), dim3(256), 0, stream, E, C, B, n, N, j0, M, D, eps_cos, eps, cos);
    return hipGetLastError();
}

// calc_loss (s3:115-127) on an explicit similarity matrix: one wave per (speaker, utterance)
// row, one workgroup per batch so the batch sum is a fixed-order reduction.
// (local rows: sim [B][n M][N], the own-speaker column of row r is j0 + r / M; n = N, j0 = 0 is the reference's calc_loss)
__global__ __launch_bounds__(256) void ge2e_calc_loss_kernel(const float* sim, int B, int n, int N, int j0, int M,
                                                             float eps, float log_eps, int variant,
                                                             float* loss, float* per) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, NW = blockDim.x >> 6;
    __shared__ float red[kMaxWaves];
    for (int bi = blockIdx.x; bi < B; bi += gridDim.x) {
        float acc = 0.f;
        for (int r = wid; r < n * M; r += NW) {
            const int j = j0 + r / M;
            const float* row = sim + ((size_t)bi * n * M + r) * N;
            const float sjj = row[j];
            float v;
            if (variant == 0) {
                float mx = -INFINITY;
                for (int k = lane; k < N; k += kWave) mx = fmaxf(mx, row[k]);
                mx = fmaxf(wave_max(mx), log_eps);
                float z = 0.f;
                for (int k = lane; k < N; k += kWave) z += expf(row[k] - mx);
                z = wave_sum(z) + expf(log_eps - mx);
                v = (mx - sjj) + logf(z);
            } else {
                float best = -INFINITY;
                for (int k = lane; k < N; k += kWave) if (k != j) best = fmaxf(best, row[k]);
                best = wave_max(best);
                const float neg = N > 1 ? 1.0f / (1.0f + expf(-best)) : 0.f;
                v = 1.0f - 1.0f / (1.0f + expf(-sjj)) + neg;
            }
            acc += v;
            if (per && lane == 0) per[(size_t)bi * n * M + r] = v;
        }
        if (lane == 0) red[wid] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float l = 0.f;
            for (int i = 0; i < NW; ++i) l += red[i];
            loss[bi] = l;
        }
        __syncthreads();
    }
}

hipError_t launch_calc_loss(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, float* loss,
                            float* per, hipStream_t stream) {
    const float log_eps = eps > 0.f ? logf(eps) : -INFINITY;
    hipLaunchKernelGGL(ge2e_calc_loss_kernel, dim3(B < 1024 ? B : 1024), dim3(256), 0, stream, sim, B, n, N, j0,
                       M, eps, log_eps, variant, loss, per);
    return hipGetLastError();
}

int generic_grid(int B) { return B < 1024 ? B : 1024; }

size_t generic_workspace_bytes(int B, int N, int M, int D) {
    return (size_t)generic_grid(B) * generic_layout(N, M, D).total * sizeof(float);
}

hipError_t launch_generic(const Problem& p, hipStream_t stream) {
    const int grid = generic_grid(p.B);
    const size_t stride = generic_layout(p.N, p.M, p.D).total;
    hipLaunchKernelGGL(ge2e_generic_kernel, dim3(grid), dim3(256), 0, stream, p, stride);
    return hipGetLastError();
}

hipError_t launch_centroids(const float* E, int B, int N, int M, int D, float* cent, hipStream_t stream) {
    const size_t total = (size_t)B * N * D;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(ge2e_centroids_kernel, dim3(grid), dim3(256), 0, stream, E, B * N, M, D, cent);
    return hipGetLastError();
}

}  // namespace ge2e
