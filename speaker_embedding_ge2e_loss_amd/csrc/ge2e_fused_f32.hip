// GE2E_IMPL_FUSED_F32: the whole loss + gradient of one (N,M,D) batch in ONE workgroup of
// ONE kernel, exact fp32, the three N*M x N x D contractions on v_mfma_f32_32x32x2_f32.
//
// Shapes: N <= 64, D in {64,128,192,256}, 2 <= M <= 64.  One 256-thread workgroup (4 waves,
// one per SIMD, the whole 512-register file each) owns a batch; the grid strides over the B
// batches of the launch.
//
//   LDS (D=256: 158 KB of the CU's 160 KB)
//     CH [64][D+4]  unit centroids c-hat (rows >= N are zero)
//     ET [64][D+4]  the current 64-row tile of unit embeddings e-hat
//     AT [64][68]   the tile's similarity rows S, then dL/dcos; reused as the epilogue's staging
//     KJ [8][D]     per-speaker constant rows of the tile (dc_j/M + sum_i du_ji/(M-1))
//     RS [64][8]    per-row scalars of the tile,  CST [64][4] per-centroid scalars
//
//   sweep 1  E (HBM)      -> speaker sums -> CH                                   (s3:34-38)
//   sweep 2  E (L2/MALL), tiles of whole speakers (<= 64 rows):
//              e-hat, leave-one-out stats (s3:96-112, s3:57)      VALU + DPP row reductions
//              X = CH . ET^T          (64 x 64 x D)               MFMA   (s3:64-70)
//              S = w (X + eps) + b, row softmax / contrast, G     VALU   (s3:27, s3:115-127)
//              gC += G_off^T . ET     (64 x D x 64 rows)          MFMA   (autograd of s3:70)
//              G_off and the row scalars are stashed in the workspace (L2)
//            gC -> through the centroid norm -> dc/M (workspace)
//   sweep 3  E (L2/MALL), same tiles:
//              gE = A' . CH           (64 rows x D x 64)          MFMA
//              dE = gE/|e| + c1 e-hat + KJ_j  -> HBM              row-layout epilogue, 16-B stores
//
// Tiles hold whole speakers so every per-speaker quantity (leave-one-out terms, dc_j) is
// tile-local.  Algebra: oracle/ge2e_oracle.py:closed_form; c1, c2, KJ are that gradient with
// u-hat = rho (s_j - e) substituted and the coefficients of e-hat, s_j collected per row; the
// s_j coefficient rides in the stashed matrix (A'[r][j] = c2 |s_j| / rne) so sweep 3's GEMM adds it.
//
// Memory pipeline: one wave per SIMD has no other wave to hide latency behind, so every
// global read is a register prefetch issued one phase early (sweep 1: a 32-row ring; sweeps
// 2/3: tile t+1 while tile t is in its MFMA phases).  All global traffic goes through buffer
// resources: one SGPR descriptor + a 32-bit lane offset (few address VGPRs), and rows outside a
// tile get an out-of-range offset, so they read 0 and their stores are dropped with no
// predication branches (a predicated load makes hipcc branch and drain vmcnt around every element).
#include "ge2e_common.hpp"
#include "ge2e_fused.hpp"

namespace ge2e {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TR = 64;      // rows per tile
constexpr int NC = 64;      // centroid slots
constexpr int APITCH = 68;  // AT row pitch (floats): 16-B aligned rows, b128 reads conflict-free
constexpr int MAX_SPT = 8;  // speakers per tile cap (KJ rows)
constexpr unsigned OOB = 0x7FFFFF00u;  // lane offset that is out of range of every buffer here

// RS / stashR columns
constexpr int R_RNE = 0, R_C1 = 1, R_C3 = 3, R_C4 = 4, R_J = 5;

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 scale4(const float4& a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int AUX = 0>  // cache-policy bits: 2 = nt (streaming)
// NOTE the offset of a 16-byte store goes entirely into the VGPR (soffset = immediate 0).  With a
// REGISTER soffset LLVM assumes the "VMEM store > 64 bit, then VALU write of its data VGPRs" hazard does
// not exist and lets the very next instruction overwrite the store's data registers; on gfx950 with two
// waves per SIMD that clobbered ~5 % of launches (4 rows x 64 columns at a time, always the younger
// wave of a SIMD).  With an immediate soffset the hazard recognizer inserts the wait state itself.
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, AUX);
}

// unit_stats with the common case (norm above the cosine eps) on v_rsq_f32 + one Newton step
// instead of sqrt and two IEEE divisions; the clamped case keeps the exact slow path.
__device__ __forceinline__ void unit_stats_fast(float sq, float eps_cos, float& rn, float& kappa) {
    if (sq > eps_cos * eps_cos && sq < 1e30f) {
        float r = __builtin_amdgcn_rsqf(sq);
        r = r * (1.5f - 0.5f * sq * r * r);
        rn = r;
        kappa = 1.0f;
    } else {
        unit_stats(sq, eps_cos, rn, kappa);
    }
}

}  // namespace

size_t fused_f32_lds_bytes(int D) {
    const int P = D + 4;
    return (size_t)(NC * P + TR * P + TR * APITCH + MAX_SPT * D + TR * 8 + NC * 4 + 16) * sizeof(float);
}

template <int NCH>  // D = 64 * NCH
__global__ __launch_bounds__(256, 1) void ge2e_fused_f32_kernel(Problem p, FusedWs wsl) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int D = 64 * NCH;
    constexpr int P = D + 4;
    constexpr unsigned ROWB = D * 4;  // bytes per embedding row
    float* const CH = smem_f;
    float* const ET = CH + NC * P;
    float* const AT = ET + TR * P;
    float* const KJ = AT + TR * APITCH;
    float* const RS = KJ + MAX_SPT * D;
    float* const CST = RS + TR * 8;
    float* const RED = CST + NC * 4;

    const int N = p.N, M = p.M, NM = N * M;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int d4 = 4 * lane;          // whole-wave row passes: this lane's 4 consecutive columns
    const bool dact = d4 < D;
    const int sub = lane >> 4;        // tile staging: 16 lanes per row, 4 rows per wave-instruction
    const int l16 = lane & 15;

    const int spt = wsl.spt;          // speakers per tile
    const int ntiles = wsl.ntiles;
    // workspace slice of this workgroup; byte offsets inside it
    const unsigned ws_bytes = (unsigned)(wsl.stride * sizeof(float));
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.ws + (size_t)blockIdx.x * wsl.stride, ws_bytes);
    const unsigned offA = (unsigned)(wsl.stash_a * 4), offR = (unsigned)(wsl.stash_rs * 4);
    const unsigned offDC = (unsigned)(wsl.dcm * 4);

    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const float fM = (float)M, inv_m = 1.0f / (float)M, inv_m1 = 1.0f / (float)(M - 1);
    const bool contrast = p.variant == 1;
    const bool want_grad = p.dE != nullptr;
    const bool slice_on = 64 * wid < D;  // this wave owns columns [64 wid, 64 wid + 64) in GEMM 2/3

    // lane offsets (bytes) of the two access shapes
    const unsigned vrow = dact ? (unsigned)d4 * 4u : OOB;                    // one row per wave
    const unsigned vtile = (unsigned)((16 * wid + sub) * D + 4 * l16) * 4u;  // 4 rows per wave, + g*4 rows

    GE2E_PROF_DECL(10)
    for (int bi = blockIdx.x; bi < p.B; bi += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rsE = make_rsrc(p.E + (size_t)bi * NM * D, (unsigned)NM * ROWB);
        const __amdgpu_buffer_rsrc_t rsG = make_rsrc(want_grad ? p.dE + (size_t)bi * NM * D : nullptr,
                                                      want_grad ? (unsigned)NM * ROWB : 0u);

        // ================= sweep 1: speaker sums -> unit centroids in LDS =================
        // Each wave streams the rows of a contiguous quarter of the speakers, one row per
        // wave-instruction, through a 32-row register ring (32 KB per wave in flight).
        {
            const int per_w = (N + 3) >> 2;
            const int jb = min(wid * per_w, N), je = min(jb + per_w, N);
            const int nr = (je - jb) * M;
            const unsigned base = (unsigned)(jb * M) * ROWB;
            constexpr int RING = 32;
            float4 ring[RING];
#pragma unroll
            for (int u = 0; u < RING; ++u)
                ring[u] = bload4(rsE, vrow, base + (unsigned)min(u, max(nr - 1, 0)) * ROWB);
            float4 s = zero4();
            int cnt = 0, j = jb;
            for (int rb = 0; rb < nr; rb += RING) {
#pragma unroll
                for (int u = 0; u < RING; ++u) {
                    const int row = rb + u;
                    if (row < nr) {
                        s.x += ring[u].x; s.y += ring[u].y; s.z += ring[u].z; s.w += ring[u].w;
                        if (++cnt == M) {
                            const float4 c = make_float4(s.x / fM, s.y / fM, s.z / fM, s.w / fM);
                            const float sq = wave_sum(dot4(c, c));
                            float rn, kap;
                            unit_stats(sq, eps_cos, rn, kap);
                            if (dact) *reinterpret_cast<float4*>(CH + j * P + d4) = scale4(c, rn);
                            if (lane == 0) {
                                CST[j * 4 + 0] = rn;          // 1 / max(|c|, eps)
                                CST[j * 4 + 1] = kap;
                                CST[j * 4 + 2] = fM / rn;     // s_j = c-hat_j * (M * max(|c|, eps))
                            }
                            s = zero4(); cnt = 0; ++j;
                        }
                    }
                    ring[u] = bload4(rsE, vrow, base + (unsigned)min(row + RING, max(nr - 1, 0)) * ROWB);
                }
            }
            for (int jz = N + wid; jz < NC; jz += 4) {        // unused centroid slots stay zero
                if (dact) *reinterpret_cast<float4*>(CH + jz * P + d4) = zero4();
                if (lane == 0) { CST[jz * 4 + 0] = 0.f; CST[jz * 4 + 1] = 0.f; CST[jz * 4 + 2] = 0.f; }
            }
        }
        __syncthreads();
        GE2E_PROF(0);

        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        f32x16 gc[2][2];  // dL/d c-hat accumulator: [k half][d half of this wave's 64-column slice]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) gc[a][b][i] = 0.f;

        // tile rows in registers: v[g][c] = row 16 wid + 4 g + sub, columns 64 c + 4 l16 .. +3;
        // rows past the tile's last speaker get the out-of-range lane offset and read 0.
        float4 v[4][NCH];
#define GE2E_LOAD_ROWS(T)                                                                   \
    do {                                                                                    \
        const int j0_ = (T) * spt;                                                          \
        const int nrows_ = min(spt, N - j0_) * M;                                           \
        const unsigned tb_ = (unsigned)(j0_ * M) * ROWB;                                    \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                     \
            const unsigned vo_ = (16 * wid + 4 * g + sub < nrows_) ? vtile : OOB;           \
            _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                 \
                v[g][c] = bload4(rsE, vo_, tb_ + (unsigned)(4 * g) * ROWB + 256u * c);      \
        }                                                                                   \
    } while (0)

        // ================= sweep 2: similarity rows, loss, dL/dcos, gC =====================
        GE2E_LOAD_ROWS(0);
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;

            // -- (a) normalise the prefetched rows, leave-one-out statistics -> ET, RS ---------
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = 16 * wid + 4 * g + sub;
                const bool rv = rl < nrows;
                const int j = j0 + (rv ? (int)(((float)rl + 0.5f) * inv_m) : 0);
                const float sc = CST[j * 4 + 2];
                float ee = 0.f, uu = 0.f, eu = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 ch = *reinterpret_cast<const float4*>(CH + j * P + 64 * c + 4 * l16);
                    const float4 e = v[g][c];
                    const float4 u = make_float4((ch.x * sc - e.x) * inv_m1, (ch.y * sc - e.y) * inv_m1,
                                                 (ch.z * sc - e.z) * inv_m1, (ch.w * sc - e.w) * inv_m1);
                    ee += dot4(e, e); uu += dot4(u, u); eu += dot4(e, u);
                }
                ee = row16_sum(ee); uu = row16_sum(uu); eu = row16_sum(eu);
                float rne, ke, rnu, ku;
                unit_stats_fast(ee, eps_cos, rne, ke);
                unit_stats_fast(uu, eps_cos, rnu, ku);
                if (!rv) rne = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    *reinterpret_cast<float4*>(ET + rl * P + 64 * c + 4 * l16) = scale4(v[g][c], rne);
                if (l16 == 0) {
                    float* rs = RS + rl * 8;
                    *reinterpret_cast<float4*>(rs) = make_float4(rne, ke, rnu, ku);
                    rs[4] = eu * rne * rnu;                 // cos(e, leave-one-out centroid)
                    rs[5] = __int_as_float(rv ? j : -1);
                }
            }
            __syncthreads();
            // prefetch the next tile (the last iteration re-requests its own tile: no branch)
            GE2E_LOAD_ROWS(min(t + 1, ntiles - 1));
            GE2E_PROF(1);

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
                    *reinterpret_cast<float4*>(Sp + 8 * g) =
                        make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
            __syncthreads();
            GE2E_PROF(2);

            // -- (c) per row: S = w (cos + eps) + b, loss, G = dL/dS -----------------------------
            // 4 lanes per row, 16 centroids per lane: row reductions are two DPP quad steps.
            {
                const int rl = 16 * wid + (lane >> 2);
                const int qk = lane & 3;
                const float4 rs0 = *reinterpret_cast<const float4*>(RS + rl * 8);  // rne ke rnu ku
                const float cosd = RS[rl * 8 + 4];
                const int j = __float_as_int(RS[rl * 8 + 5]);
                const bool rv = j >= 0;
                const float rne = rs0.x, ke = rs0.y, rnu = rs0.z, ku = rs0.w;
                float c0[16];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 t4 = *reinterpret_cast<const float4*>(AT + rl * APITCH + 16 * qk + 4 * i);
                    c0[4 * i] = t4.x; c0[4 * i + 1] = t4.y; c0[4 * i + 2] = t4.z; c0[4 * i + 3] = t4.w;
                }
                const int jrel = j - 16 * qk;  // own-speaker column relative to this lane's 16
#pragma unroll
                for (int i = 0; i < 16; ++i) if (i == jrel) c0[i] = cosd;
                const float sjj = w * (cosd + eps) + bias;
                float sv[16], g[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) sv[i] = (16 * qk + i < N) ? w * (c0[i] + eps) + bias : -INFINITY;
                float per;
                if (!contrast) {
                    float mx = sv[0];
#pragma unroll
                    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sv[i]);
                    mx = fmaxf(quad_max(mx), log_eps);
                    float zoff = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        g[i] = __expf(sv[i] - mx);  // exp(-inf) = 0 for padded centroids
                        if (i != jrel) zoff += g[i];
                    }
                    zoff = quad_sum(zoff) + __expf(log_eps - mx);
                    const float z = zoff + __expf(sjj - mx);
                    per = (mx - sjj) + __logf(z);
                    const float rz = 1.0f / z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) g[i] = (i == jrel) ? -zoff * rz : g[i] * rz;  // 1 - p_jj = z_off / z
                } else {
                    float best = -INFINITY; int besti = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (i != jrel && sv[i] > best) { best = sv[i]; besti = 16 * qk + i; }
                    quad_argmax(best, besti);
                    const float pos = 1.0f / (1.0f + __expf(-sjj));
                    const float neg = (N > 1) ? 1.0f / (1.0f + __expf(-best)) : 0.0f;
                    per = 1.0f - pos + neg;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        g[i] = (i == jrel) ? -pos * (1.0f - pos) : ((16 * qk + i == besti) ? neg * (1.0f - neg) : 0.f);
                }
                float coef = 0.f, ad = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (!rv || 16 * qk + i >= N) g[i] = 0.f;
                    dw_acc += g[i] * (c0[i] + eps);
                    db_acc += g[i];
                    const float av = w * g[i];
                    coef += av * c0[i];             // (dL/d e-hat) . e-hat, own-speaker term included
                    if (i == jrel) { ad = av; g[i] = 0.f; } else { g[i] = av; }
                }
                coef = quad_sum(coef);
                ad = quad_sum(ad);                  // dL/dcos on the own-speaker column
                const float rne1 = rv ? rne : 1.0f;
                const float rho = rnu * inv_m1;
                const float c2 = rho * (ad * rne1 + ad * ku * cosd * rnu * inv_m1);
                const float fold = rv ? c2 * CST[(rv ? j : 0) * 4 + 2] / rne1 : 0.f;
                const unsigned va = (unsigned)(((t * TR + rl) * NC + 16 * qk) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 o4 = make_float4(g[4 * i], g[4 * i + 1], g[4 * i + 2], g[4 * i + 3]);
                    *reinterpret_cast<float4*>(AT + rl * APITCH + 16 * qk + 4 * i) = o4;
                    if (4 * i == (jrel & ~3)) {
                        if ((jrel & 3) == 0) o4.x = fold; else if ((jrel & 3) == 1) o4.y = fold;
                        else if ((jrel & 3) == 2) o4.z = fold; else o4.w = fold;
                    }
                    bstore4(rsW, va, offA + 16u * i, o4);     // stash A' (unconditional: countable by vmcnt)
                }
                if (rv && qk == 0) {
                    loss_acc += per;
                    if (p.per) p.per[(size_t)bi * NM + r0 + rl] = per;
                }
                {
                    const float c1 = (-ke * coef * rne1 - ad * rnu * inv_m1) - c2 / rne1;
                    const float alpha = ad * rnu * (1.0f + ku * cosd * rho / rne1);
                    const float beta = -ad * rnu * ku * cosd * rho;
                    const unsigned vr = (qk == 0) ? (unsigned)((t * TR + rl) * 32) : OOB;
                    bstore4(rsW, vr, offR, make_float4(rne, c1, 0.f, alpha * inv_m1));
                    bstore4(rsW, vr, offR + 16u, make_float4(beta * inv_m1, __int_as_float(j), 0.f, 0.f));
                }
            }
            __syncthreads();
            GE2E_PROF(3);

            // -- (d) gC[k][d] += sum_r A_off[r][k] ET[r][d]; wave owns a 64-column slice of d -----
            if (slice_on && want_grad) {
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
            GE2E_PROF(4);
        }

        // ---- batch scalars: fixed-order reduction over the 4 waves ---------------------------
        loss_acc = wave_sum(loss_acc);
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
            float4 g = zero4(), c = g;
            if (dact) {
                g = *reinterpret_cast<const float4*>(ET + k * P + d4);
                c = *reinterpret_cast<const float4*>(CH + k * P + d4);
            }
            const float coef = wave_sum(dot4(g, c));
            const float rn = CST[k * 4 + 0], kap = CST[k * 4 + 1];
            const float f = kap * coef, sc = rn / fM;
            bstore4(rsW, vrow, offDC + (unsigned)k * ROWB,
                    make_float4((g.x - f * c.x) * sc, (g.y - f * c.y) * sc, (g.z - f * c.z) * sc, (g.w - f * c.w) * sc));
        }
        __syncthreads();
        GE2E_PROF(5);

        // ================= sweep 3: gE = A' . CH, epilogue -> dE ===========================
        // prefetch group of a tile: dc rows of its speakers (this wave's: j0 + wid, j0 + wid + 4),
        // the e rows, the stashed A' tile and row scalars.  (macro: arrays captured by a lambda land
        // in scratch memory with a vmcnt(0) drain per element)
        float4 a4_0, a4_1, a4_2, a4_3, r4, dcm_0, dcm_1;
        float rne_0, rne_1, rne_2, rne_3;
#define GE2E_LOAD_TILE3(T)                                                                            \
    do {                                                                                              \
        const int t_ = (T);                                                                           \
        dcm_0 = bload4(rsW, vrow, offDC + (unsigned)min(t_ * spt + wid, N - 1) * ROWB);               \
        dcm_1 = bload4(rsW, vrow, offDC + (unsigned)min(t_ * spt + wid + 4, N - 1) * ROWB);           \
        GE2E_LOAD_ROWS(t_);                                                                           \
        const unsigned ta_ = offA + (unsigned)t_ * (TR * NC * 4);                                     \
        a4_0 = bload4(rsW, (unsigned)tid * 16u, ta_);                                                 \
        a4_1 = bload4(rsW, (unsigned)tid * 16u, ta_ + 4096u);                                         \
        a4_2 = bload4(rsW, (unsigned)tid * 16u, ta_ + 8192u);                                         \
        a4_3 = bload4(rsW, (unsigned)tid * 16u, ta_ + 12288u);                                        \
        const unsigned tr_ = offR + (unsigned)t_ * (TR * 32);                                         \
        r4 = bload4(rsW, (unsigned)(tid & 127) * 16u, tr_);                                           \
        rne_0 = bload1(rsW, (unsigned)(16 * wid + sub) * 32u, tr_);                                   \
        rne_1 = bload1(rsW, (unsigned)(16 * wid + sub) * 32u, tr_ + 128u);                            \
        rne_2 = bload1(rsW, (unsigned)(16 * wid + sub) * 32u, tr_ + 256u);                            \
        rne_3 = bload1(rsW, (unsigned)(16 * wid + sub) * 32u, tr_ + 384u);                            \
    } while (0)

        GE2E_LOAD_TILE3(0);
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;
            // -- (a) stage the prefetched tile: e-hat, A', row scalars ---------------------------
            {
                if (tid < TR * 2) reinterpret_cast<float4*>(RS)[tid] = r4;
                const int r = tid >> 4, c4 = (tid & 15) * 4;   // float4 index tid + 256 i -> row r + 16 i
                *reinterpret_cast<float4*>(AT + r * APITCH + c4) = a4_0;
                *reinterpret_cast<float4*>(AT + (r + 16) * APITCH + c4) = a4_1;
                *reinterpret_cast<float4*>(AT + (r + 32) * APITCH + c4) = a4_2;
                *reinterpret_cast<float4*>(AT + (r + 48) * APITCH + c4) = a4_3;
                const float rne_g[4] = {rne_0, rne_1, rne_2, rne_3};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rl = 16 * wid + 4 * g + sub;
#pragma unroll
                    for (int c = 0; c < NCH; ++c)  // pad rows: v = 0 and rne = 0
                        *reinterpret_cast<float4*>(ET + rl * P + 64 * c + 4 * l16) = scale4(v[g][c], rne_g[g]);
                }
            }
            const float4 dcm_cur[2] = {dcm_0, dcm_1};
            __syncthreads();
            GE2E_LOAD_TILE3(min(t + 1, ntiles - 1));
            GE2E_PROF(6);
            // -- (b) per-speaker constant rows KJ_j = dc_j/M + sum_i (c3_i e-hat_i + c4_i s_j) ----
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                const int jl = wid + 4 * jq;
                const int j = j0 + jl;
                if (dact && jl < nspk) {
                    float4 acc = dcm_cur[jq];
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
            // -- (c) gE[r][d] = sum_k A'[r][k] CH[k][d]; wave owns its 64-column slice ----------
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
            __syncthreads();  // KJ complete, every wave is done reading AT
            GE2E_PROF(7);
            // -- (d) epilogue: accumulator layout (lane = column) -> row layout through this wave's
            //        16 x 64 staging block (AT is dead now), then 16-byte stores, 4 rows x 256 B
            //        per wave-instruction:  dE = gE' rne + c1 e-hat + KJ_j
            if (slice_on) {
                float* ST = AT + wid * (16 * APITCH);
                const unsigned vst = (unsigned)(sub * D + 64 * wid + 4 * l16) * 4u;
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) {      // four chunks of 16 rows
                    const int a = cq >> 1, gb = 2 * (cq & 1);
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                ST[(8 * gg + 4 * h + q) * APITCH + 32 * b + l31] = ge[a][b][4 * (gb + gg) + q];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int rloc = 4 * ps + sub;
                        const int rl = 16 * cq + rloc;
                        const bool rv = rl < nrows;
                        const float4 acc = *reinterpret_cast<const float4*>(ST + rloc * APITCH + 4 * l16);
                        const float rne = RS[rl * 8 + R_RNE], c1 = RS[rl * 8 + R_C1];
                        const int jl = rv ? __float_as_int(RS[rl * 8 + R_J]) - j0 : 0;
                        const float4 e = *reinterpret_cast<const float4*>(ET + rl * P + 64 * wid + 4 * l16);
                        const float4 kj = *reinterpret_cast<const float4*>(KJ + jl * D + 64 * wid + 4 * l16);
                        // pad rows get an out-of-range offset: the store is dropped, no branch
                        bstore4<2>(rsG, rv ? vst : OOB, (unsigned)(r0 + 16 * cq + 4 * ps) * ROWB,
                                make_float4(acc.x * rne + e.x * c1 + kj.x, acc.y * rne + e.y * c1 + kj.y,
                                            acc.z * rne + e.z * c1 + kj.z, acc.w * rne + e.w * c1 + kj.w));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            GE2E_PROF(8);
        }
    }
    GE2E_PROF_FLUSH(10)
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
    L.dump = L.dcm + (size_t)NC * D;
    L.stride = align_up(L.dump, 64);
    return L;
}

int fused_f32_grid(int B) { return B < 256 ? B : 256; }

size_t fused_f32_workspace_bytes(int B, int N, int M, int D) {
    return (size_t)fused_f32_grid(B) * fused_f32_layout(N, M, D).stride * sizeof(float);
}

template <int NCH>
static hipError_t launch_nch(const Problem& p, const FusedWs& L, size_t lds, hipStream_t stream) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_fused_f32_kernel<NCH>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(ge2e_fused_f32_kernel<NCH>, dim3(fused_f32_grid(p.B)), dim3(256), lds, stream, p, L);
    return hipGetLastError();
}

hipError_t launch_fused_f32(const Problem& p, hipStream_t stream) {
    const size_t lds = fused_f32_lds_bytes(p.D);
    const FusedWs L = fused_f32_layout(p.N, p.M, p.D);
    switch (p.D / 64) {
        case 1: return launch_nch<1>(p, L, lds, stream);
        case 2: return launch_nch<2>(p, L, lds, stream);
        case 3: return launch_nch<3>(p, L, lds, stream);
        default: return launch_nch<4>(p, L, lds, stream);
    }
}

}  // namespace ge2e
