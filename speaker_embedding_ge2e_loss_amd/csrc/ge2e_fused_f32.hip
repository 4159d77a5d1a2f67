// GE2E_IMPL_FUSED_F32: the whole loss + gradient of one (N,M,D) batch in ONE workgroup of
// ONE kernel, exact fp32, the three N*M x N x D contractions on v_mfma_f32_32x32x2_f32.
//
// Shapes: N <= 64, D in {64,128,192,256}, 2 <= M <= 64.  One 256-thread workgroup (4 waves,
// one per SIMD) owns a batch; the grid strides over the B batches of the launch.
//
//   LDS (D=256: 158 KB of the CU's 160 KB)
//     CH [64][D+4]  unit centroids c-hat (rows >= N are zero)
//     ET [64][D+4]  the current 64-row tile of unit embeddings e-hat
//     AT [64][68]   the tile's similarity rows S, then dL/dcos (own-speaker column zeroed)
//     KJ [8][D]     per-speaker constant rows of the tile (dc_j/M + sum_i du_ji/(M-1))
//     RS [64][8]    per-row scalars of the tile,  CST [64][4] per-centroid scalars
//
//   sweep 1  E (HBM)      -> speaker sums -> CH                                   (s3:34-38)
//   sweep 2  E (L2/MALL), tiles of whole speakers (<= 64 rows):
//              e-hat, leave-one-out stats (s3:96-112, s3:57)      VALU + wave reductions
//              X = CH . ET^T          (64 x 64 x D)               MFMA   (s3:64-70)
//              S = w (X + eps) + b, row softmax / contrast, G     VALU   (s3:27, s3:115-127)
//              gC += G_off^T . ET     (64 x D x 64 rows)          MFMA   (autograd of s3:70)
//              G_off and the row scalars are stashed in the workspace (L2)
//            gC -> through the centroid norm -> dc/M (workspace)
//   sweep 3  E (L2/MALL), same tiles:
//              gE = G_off . CH        (64 rows x D x 64)          MFMA
//              dE = gE/|e| + c1 e-hat + c2 s_j + KJ_j  -> HBM     VALU epilogue
//
// Tiles hold whole speakers so every per-speaker quantity (leave-one-out terms, dc_j) is
// tile-local.  Algebra: oracle/ge2e_oracle.py:closed_form; c1, c2, KJ are that gradient with
// u-hat = rho (s_j - e) substituted and the coefficients of e-hat, s_j collected per row.
#include "ge2e_common.hpp"
#include "ge2e_fused.hpp"

namespace ge2e {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TR = 64;      // rows per tile
constexpr int NC = 64;      // centroid slots
constexpr int APITCH = 68;  // AT row pitch (floats): 16-B aligned rows, b128 reads conflict-free
constexpr int MAX_SPT = 8;  // speakers per tile cap (KJ rows)

// RS columns
constexpr int R_RNE = 0, R_C1 = 1, R_C2S = 2, R_C3 = 3, R_C4 = 4, R_J = 5, R_KE = 6, R_X = 7;

struct Smem {
    float *CH, *ET, *AT, *KJ, *RS, *CST, *RED;
};

__device__ __forceinline__ Smem carve(float* base, int D) {
    const int P = D + 4;
    Smem s;
    s.CH = base;
    s.ET = s.CH + NC * P;
    s.AT = s.ET + TR * P;
    s.KJ = s.AT + TR * APITCH;
    s.RS = s.KJ + MAX_SPT * D;
    s.CST = s.RS + TR * 8;
    s.RED = s.CST + NC * 4;
    return s;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

}  // namespace

size_t fused_f32_lds_bytes(int D) {
    const int P = D + 4;
    return (size_t)(NC * P + TR * P + TR * APITCH + MAX_SPT * D + TR * 8 + NC * 4 + 16) * sizeof(float);
}

__global__ __launch_bounds__(256, 1) void ge2e_fused_f32_kernel(Problem p, FusedWs wsl) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int N = p.N, M = p.M, D = p.D, NM = N * M;
    const int P = D + 4;
    const Smem sm = carve(smem_f, D);
    float* const CH = sm.CH; float* const ET = sm.ET; float* const AT = sm.AT;
    float* const KJ = sm.KJ; float* const RS = sm.RS; float* const CST = sm.CST; float* const RED = sm.RED;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int d4 = 4 * lane;          // this lane's 4 consecutive columns in row-wise passes
    const bool dact = d4 < D;

    const int spt = wsl.spt;          // speakers per tile
    const int ntiles = wsl.ntiles;
    float* const ws = p.ws + (size_t)blockIdx.x * wsl.stride;
    float* const stashA = ws + wsl.stash_a;    // [ntiles][64][64]
    float* const stashR = ws + wsl.stash_rs;   // [ntiles][64][8]
    float* const DCM = ws + wsl.dcm;           // [64][D]   dL/dc / M

    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const float fM = (float)M, inv_m1 = 1.0f / (float)(M - 1);
    const bool contrast = p.variant == 1;
    const bool want_grad = p.dE != nullptr;

    for (int bi = blockIdx.x; bi < p.B; bi += gridDim.x) {
        const float* __restrict__ E = p.E + (size_t)bi * NM * D;

        // ================= sweep 1: speaker sums -> unit centroids in LDS =================
        for (int j = wid; j < NC; j += 4) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < N && dact) {
                const float* base = E + (size_t)j * M * D + d4;
                for (int i0 = 0; i0 < M; i0 += 8) {
                    float4 v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        v[i] = (i0 + i < M) ? *reinterpret_cast<const float4*>(base + (size_t)(i0 + i) * D)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
                }
            }
            float4 c = make_float4(s.x / fM, s.y / fM, s.z / fM, s.w / fM);
            const float sq = wave_sum(dot4(c, c));
            float rn, kap;
            unit_stats(sq, eps_cos, rn, kap);
            if (dact) *reinterpret_cast<float4*>(CH + j * P + d4) = make_float4(c.x * rn, c.y * rn, c.z * rn, c.w * rn);
            if (lane == 0) {
                CST[j * 4 + 0] = rn;          // 1 / max(|c|, eps)
                CST[j * 4 + 1] = kap;
                CST[j * 4 + 2] = fM / rn;     // s_j = c-hat_j * (M * max(|c|, eps))
            }
        }
        __syncthreads();

        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        f32x16 gc[2][2];  // dL/d c-hat accumulator: [k half][d half of this wave's 64-column slice]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) gc[a][b][i] = 0.f;
        const bool slice_on = 64 * wid < D;  // this wave owns columns [64 wid, 64 wid + 64)

        // ================= sweep 2: similarity rows, loss, dL/dcos, gC =====================
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;

            // -- (a) load 16 rows per wave, normalise, leave-one-out statistics --------------
            {
                float4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rl = 16 * wid + i;
                    v[i] = (rl < nrows && dact) ? *reinterpret_cast<const float4*>(E + (size_t)(r0 + rl) * D + d4)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rl = 16 * wid + i;
                    const bool rv = rl < nrows;
                    const int jl = rv ? rl / M : 0;
                    const int j = j0 + jl;
                    const float sc = CST[j * 4 + 2];
                    float4 c = dact ? *reinterpret_cast<const float4*>(CH + j * P + d4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 e = v[i];
                    float4 u = make_float4((c.x * sc - e.x) * inv_m1, (c.y * sc - e.y) * inv_m1,
                                           (c.z * sc - e.z) * inv_m1, (c.w * sc - e.w) * inv_m1);
                    float ee = dot4(e, e), uu = dot4(u, u), eu = dot4(e, u);
                    ee = wave_sum(ee); uu = wave_sum(uu); eu = wave_sum(eu);
                    float rne, ke, rnu, ku;
                    unit_stats(ee, eps_cos, rne, ke);
                    unit_stats(uu, eps_cos, rnu, ku);
                    if (dact)
                        *reinterpret_cast<float4*>(ET + rl * P + d4) =
                            rv ? make_float4(e.x * rne, e.y * rne, e.z * rne, e.w * rne) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (lane == 0) {
                        float* rs = RS + rl * 8;
                        rs[0] = rne; rs[1] = ke; rs[2] = rnu; rs[3] = ku;
                        rs[4] = eu * rne * rnu;                 // cos(e, leave-one-out centroid)
                        rs[5] = __int_as_float(rv ? j : -1);
                    }
                }
            }
            __syncthreads();

            // -- (b) X[k][r] = sum_d CH[k][d] ET[r][d]; wave (a,b) owns k-half a, r-half b ------
            {
                const int a = wid >> 1, b = wid & 1;
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const float* Ap = CH + (32 * a + l31) * P + 4 * h;
                const float* Bp = ET + (32 * b + l31) * P + 4 * h;
                // K order inside a group of 8: lane half h supplies d = 8q + 4h + i at step i
#pragma unroll 4
                for (int q = 0; q < D / 8; ++q) {
                    const float4 av = *reinterpret_cast<const float4*>(Ap + 8 * q);
                    const float4 bv = *reinterpret_cast<const float4*>(Bp + 8 * q);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
                }
                // C layout: column (r) = lane&31, row (k) = (reg&3) + 8 (reg>>2) + 4 h
                float* Sp = AT + (32 * b + l31) * APITCH + 32 * a + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(Sp + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
            __syncthreads();

            // -- (c) per row: S = w (cos + eps) + b, loss, G = dL/dS; lane <-> centroid k -------
            for (int i = 0; i < 16; ++i) {
                const int rl = 16 * wid + i;
                const float* rs = RS + rl * 8;
                const int j = __float_as_int(rs[5]);
                const bool rv = j >= 0;
                const float rne = rs[0], ke = rs[1], rnu = rs[2], ku = rs[3], cosd = rs[4];
                const int k = lane;
                const bool kv = k < N;
                const float c0 = (k == j) ? cosd : AT[rl * APITCH + k];
                const float s = kv ? w * (c0 + eps) + bias : -INFINITY;
                const float sjj = w * (cosd + eps) + bias;
                float g, per;
                if (!contrast) {
                    const float mx = fmaxf(wave_max(s), log_eps);
                    const float ex = expf(s - mx);  // exp(-inf) = 0 for padded centroids
                    const float zoff = wave_sum(k == j ? 0.f : ex) + expf(log_eps - mx);
                    const float z = zoff + expf(sjj - mx);
                    per = (mx - sjj) + logf(z);
                    const float rz = 1.0f / z;
                    g = (k == j) ? -zoff * rz : ex * rz;  // 1 - p_jj = z_off / z: no cancellation
                } else {
                    float best = (kv && k != j) ? s : -INFINITY;
                    int besti = (kv && k != j) ? k : 0x7fffffff;
                    wave_argmax(best, besti);
                    const float pos = 1.0f / (1.0f + expf(-sjj));
                    const float neg = (N > 1) ? 1.0f / (1.0f + expf(-best)) : 0.0f;
                    per = 1.0f - pos + neg;
                    g = (k == j) ? -pos * (1.0f - pos) : ((k == besti) ? neg * (1.0f - neg) : 0.f);
                }
                if (!rv || !kv) g = 0.f;
                dw_acc += g * (c0 + eps);
                db_acc += g;
                const float av = w * g;
                const float coef = wave_sum(av * c0);        // (dL/d e-hat) . e-hat
                const float ad = wave_sum(k == j ? av : 0.f); // dL/dcos on the own-speaker column
                const float aoff = (k == j) ? 0.f : av;
                AT[rl * APITCH + k] = aoff;
                if (want_grad) stashA[((size_t)t * TR + rl) * NC + k] = aoff;
                if (rv) {
                    loss_acc += per;
                    if (p.per && lane == 0) p.per[(size_t)bi * NM + r0 + rl] = per;
                }
                if (want_grad && lane == 0) {
                    // dE_r = gE_off rne + c1 e-hat + c2 s_j + KJ_j   (header comment)
                    const float rho = rnu * inv_m1;
                    const float c2 = rho * (ad * rne + ad * ku * cosd * rnu * inv_m1);
                    const float c1 = (-ke * coef * rne - ad * rnu * inv_m1) - c2 / rne;
                    const float alpha = ad * rnu * (1.0f + ku * cosd * rho / rne);
                    const float beta = -ad * rnu * ku * cosd * rho;
                    float* o = stashR + ((size_t)t * TR + rl) * 8;
                    o[R_RNE] = rne;
                    o[R_C1] = c1;
                    o[R_C2S] = rv ? c2 * CST[j * 4 + 2] : 0.f;
                    o[R_C3] = alpha * inv_m1;
                    o[R_C4] = beta * inv_m1;
                    o[R_J] = __int_as_float(j);
                    o[R_KE] = 0.f; o[R_X] = 0.f;
                }
            }
            if (!want_grad) { __syncthreads(); continue; }
            __syncthreads();

            // -- (d) gC[k][d] += sum_r A_off[r][k] ET[r][d]; wave owns a 64-column slice of d -----
            if (slice_on) {
                const float* Ap = AT + h * APITCH + l31;
                const float* Bp = ET + h * P + 64 * wid + l31;
#pragma unroll 4
                for (int t2 = 0; t2 < TR / 2; ++t2) {
                    const float a0 = Ap[2 * t2 * APITCH], a1 = Ap[2 * t2 * APITCH + 32];
                    const float b0 = Bp[2 * t2 * P], b1 = Bp[2 * t2 * P + 32];
                    gc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, gc[0][0], 0, 0, 0);
                    gc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, gc[0][1], 0, 0, 0);
                    gc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, gc[1][0], 0, 0, 0);
                    gc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, gc[1][1], 0, 0, 0);
                }
            }
            __syncthreads();
        }

        // ---- batch scalars: fixed-order reduction over the 4 waves ---------------------------
        dw_acc = wave_sum(dw_acc);
        db_acc = wave_sum(db_acc);
        if (lane == 0) { RED[wid] = loss_acc; RED[4 + wid] = dw_acc; RED[8 + wid] = db_acc; }
        __syncthreads();
        if (tid == 0) {
            if (p.loss) p.loss[bi] = (RED[0] + RED[1]) + (RED[2] + RED[3]);
            if (p.dw) p.dw[bi] = (RED[4] + RED[5]) + (RED[6] + RED[7]);
            if (p.db) p.db[bi] = (RED[8] + RED[9]) + (RED[10] + RED[11]);
        }
        if (!want_grad) { __syncthreads(); continue; }

        // ---- gC -> LDS (reusing ET) -> through the centroid norm -> dc / M in the workspace ----
        if (slice_on) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int k = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                        ET[k * P + 64 * wid + 32 * b + l31] = gc[a][b][i];
                    }
        }
        __syncthreads();
        for (int k = wid; k < N; k += 4) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f), c = g;
            if (dact) {
                g = *reinterpret_cast<const float4*>(ET + k * P + d4);
                c = *reinterpret_cast<const float4*>(CH + k * P + d4);
            }
            const float coef = wave_sum(dot4(g, c));
            const float rn = CST[k * 4 + 0], kap = CST[k * 4 + 1];
            const float f = kap * coef, sc = rn / fM;
            if (dact)
                *reinterpret_cast<float4*>(DCM + k * D + d4) =
                    make_float4((g.x - f * c.x) * sc, (g.y - f * c.y) * sc, (g.z - f * c.z) * sc, (g.w - f * c.w) * sc);
        }
        __syncthreads();

        // ================= sweep 3: gE = A_off . CH, epilogue -> dE ===========================
        float* __restrict__ dE = p.dE + (size_t)bi * NM * D;
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;
            // -- (a) stage the tile: e-hat, A_off, row scalars ---------------------------------
            {
                const float4* rsrc = reinterpret_cast<const float4*>(stashR + (size_t)t * TR * 8);
                if (tid < TR * 2) reinterpret_cast<float4*>(RS)[tid] = rsrc[tid];
                const float4* asrc = reinterpret_cast<const float4*>(stashA + (size_t)t * TR * NC);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = tid + 256 * i;           // float4 index in the [64][64] tile
                    const int r = idx >> 4, c4 = (idx & 15) * 4;
                    *reinterpret_cast<float4*>(AT + r * APITCH + c4) = asrc[idx];
                }
                float4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rl = 16 * wid + i;
                    v[i] = (rl < nrows && dact) ? *reinterpret_cast<const float4*>(E + (size_t)(r0 + rl) * D + d4)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rl = 16 * wid + i;
                    const float rne = stashR[((size_t)t * TR + rl) * 8 + R_RNE];
                    if (dact)
                        *reinterpret_cast<float4*>(ET + rl * P + d4) =
                            make_float4(v[i].x * rne, v[i].y * rne, v[i].z * rne, v[i].w * rne);
                }
            }
            __syncthreads();
            // -- (b) per-speaker constant rows KJ_j = dc_j/M + sum_i (c3_i e-hat_i + c4_i s_j) ----
            for (int jl = wid; jl < nspk; jl += 4) {
                const int j = j0 + jl;
                if (dact) {
                    float4 acc = *reinterpret_cast<const float4*>(DCM + j * D + d4);
                    float bsum = 0.f;
                    for (int i = 0; i < M; ++i) {
                        const int rl = jl * M + i;
                        const float c3 = RS[rl * 8 + R_C3];
                        bsum += RS[rl * 8 + R_C4];
                        const float4 e = *reinterpret_cast<const float4*>(ET + rl * P + d4);
                        acc.x += c3 * e.x; acc.y += c3 * e.y; acc.z += c3 * e.z; acc.w += c3 * e.w;
                    }
                    const float4 c = *reinterpret_cast<const float4*>(CH + j * P + d4);
                    const float bs = bsum * CST[j * 4 + 2];
                    *reinterpret_cast<float4*>(KJ + jl * D + d4) =
                        make_float4(acc.x + bs * c.x, acc.y + bs * c.y, acc.z + bs * c.z, acc.w + bs * c.w);
                }
            }
            // -- (c) gE[r][d] = sum_k A_off[r][k] CH[k][d]; wave owns its 64-column slice -------
            f32x16 ge[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) ge[a][b][i] = 0.f;
            if (slice_on) {
                const float* Ap = AT + l31 * APITCH + 4 * h;
                const float* Bp = CH + 4 * h * P + 64 * wid + l31;
#pragma unroll 2
                for (int q = 0; q < NC / 8; ++q) {
                    const float4 a0 = *reinterpret_cast<const float4*>(Ap + 8 * q);
                    const float4 a1 = *reinterpret_cast<const float4*>(Ap + 32 * APITCH + 8 * q);
                    const float av0[4] = {a0.x, a0.y, a0.z, a0.w};
                    const float av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float b0 = Bp[(8 * q + i) * P], b1 = Bp[(8 * q + i) * P + 32];
                        ge[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[i], b0, ge[0][0], 0, 0, 0);
                        ge[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[i], b1, ge[0][1], 0, 0, 0);
                        ge[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[i], b0, ge[1][0], 0, 0, 0);
                        ge[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[i], b1, ge[1][1], 0, 0, 0);
                    }
                }
            }
            __syncthreads();  // KJ complete
            // -- (d) epilogue straight from the accumulator layout --------------------------------
            if (slice_on) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int rl = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
                        if (rl < nrows) {
                            const float4 rs = *reinterpret_cast<const float4*>(RS + rl * 8);  // rne, c1, c2s, c3
                            const int j = __float_as_int(RS[rl * 8 + R_J]);
                            const int jl = j - j0;
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                const int d = 64 * wid + 32 * b + l31;
                                const float v = ge[a][b][i] * rs.x + ET[rl * P + d] * rs.y + CH[j * P + d] * rs.z +
                                                KJ[jl * D + d];
                                dE[(size_t)(r0 + rl) * D + d] = v;
                            }
                        }
                    }
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------
bool fused_f32_supports(int N, int M, int D) {
    return N >= 1 && N <= 64 && M >= 2 && M <= 64 && D >= 64 && D <= 256 && (D % 64) == 0;
}

FusedWs fused_f32_layout(int N, int M, int D) {
    FusedWs L;
    int spt = TR / M;
    if (spt > MAX_SPT) spt = MAX_SPT;
    if (spt > N) spt = N;
    L.spt = spt;
    L.ntiles = (N + spt - 1) / spt;
    L.stash_a = 0;
    L.stash_rs = L.stash_a + (size_t)L.ntiles * TR * NC;
    L.dcm = L.stash_rs + (size_t)L.ntiles * TR * 8;
    L.stride = align_up(L.dcm + (size_t)NC * D, 64);
    return L;
}

int fused_f32_grid(int B) { return B < 256 ? B : 256; }

size_t fused_f32_workspace_bytes(int B, int N, int M, int D) {
    return (size_t)fused_f32_grid(B) * fused_f32_layout(N, M, D).stride * sizeof(float);
}

hipError_t launch_fused_f32(const Problem& p, hipStream_t stream) {
    const size_t lds = fused_f32_lds_bytes(p.D);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_fused_f32_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    const FusedWs L = fused_f32_layout(p.N, p.M, p.D);
    hipLaunchKernelGGL(ge2e_fused_f32_kernel, dim3(fused_f32_grid(p.B)), dim3(256), lds, stream, p, L);
    return hipGetLastError();
}

}  // namespace ge2e
