// GE2E_IMPL_FUSED_SPLIT's per-workgroup body (one 512-thread workgroup per batch, three sweeps over E, split-fp16 MFMA) as a
// device function: ge2e_fused_split.hip wraps it in a kernel of its own; the team kernels call it when a launch has to be
// redone without teams (no team formed, a hand-off timed out, an untrusted control block) -- inside the SAME launch, where
// rounds 2-4 queued a second, gated launch behind every team call.  Everything lives in namespace ge2e::fsplit (its helpers
// have the names of the team kernels' own).
#pragma once
#include "ge2e_common.hpp"
#include "ge2e_fused.hpp"
#include "ge2e_split_gemm.hpp"

namespace ge2e {
namespace fsplit {


typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TR = 64;      // rows per tile
constexpr int NC = 64;      // centroid slots
constexpr int APITCH = 68;  // S / staging row pitch (floats)
constexpr int GP = 72;      // G image row pitch (halfs)
constexpr int MAX_SPT = 6;  // speakers per tile cap (KJ rows; 8 would not fit the LDS budget)
constexpr int NWAVE = 8;
constexpr unsigned OOB = 0x7FFFFF00u;  // lane offset that is out of range of every buffer here

// RS / stashR columns

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
// cache-policy bits of buffer instructions on gfx950: 1 = sc0, 2 = nt (streaming), 16 = sc1
#ifndef GE2E_AUX_E1
#define GE2E_AUX_E1 0   // sweep-1 read of E (first touch, re-read twice later)
#endif
#ifndef GE2E_AUX_E3
#define GE2E_AUX_E3 2   // sweep-3 read of E (last use): nt, +0.5 % measured
#endif
#ifndef GE2E_AUX_DE
#define GE2E_AUX_DE 2   // dE stores (never re-read here): nt keeps E resident for the re-reads, +4.5 % measured
#endif
template <int AUX = 0>
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int AUX = 0>
// NOTE the offset of a 16-byte store goes entirely into the VGPR (soffset = immediate 0).  With a
// REGISTER soffset LLVM assumes the "VMEM store > 64 bit, then VALU write of its data VGPRs" hazard does
// not exist and lets the very next instruction overwrite the store's data registers; on gfx950 with two
// waves per SIMD that clobbered ~5 % of launches (4 rows x 64 columns at a time, always the younger
// wave of a SIMD).  With an immediate soffset the hazard recognizer inserts the wait state itself.
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff + soff, 0, AUX);
}
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
// write 4 scaled values as fp16 hi / lo at the same (row, col) of two images
__device__ __forceinline__ void put_split4(_Float16* hi_img, _Float16* lo_img, int off, const float4& x) {
    h4 hi, lo;
    split4(x, hi, lo);
    *reinterpret_cast<h4*>(hi_img + off) = hi;
    *reinterpret_cast<h4*>(lo_img + off) = lo;
}
__device__ __forceinline__ float4 get_join4(const _Float16* hi_img, const _Float16* lo_img, int off) {
    return join4(*reinterpret_cast<const h4*>(hi_img + off), *reinterpret_cast<const h4*>(lo_img + off));
}


// One 512-thread workgroup works through the batches wg, wg + nwg, ... of the call; `smem_f` = fused_split_lds_bytes(D) of LDS,
// workspace slice `wg` of p.ws.  Called by the kernel of its own (ge2e_fused_split.hip) and by the team kernels when a launch
// has to be redone by one workgroup per batch (ge2e_team.hip, ge2e_team_fwd.hip).
template <int NCH>  // D = 64 * NCH
__device__ __forceinline__ void body(const Problem& p, const FusedWs& wsl, float* const smem_f, const int wg, const int nwg) {
    constexpr int D = 64 * NCH;
    constexpr int P = D + 4;    // fp32 pitch of the gC staging that reuses the ET images in finalize
    constexpr int PH = D + 16;  // fp16 image pitch: rows 8 banks apart (mod 64) -> neither the b128 row reads nor the
                                // 4-row x 32-byte transposing reads collide (D + 8 cost 39 % of the LDS cycles in conflicts)
    constexpr unsigned ROWB = D * 4;  // bytes per row of the WORKSPACE and LDS layouts (D = 64 NCH columns)
    // The caller's D may be smaller than 64 NCH (any multiple of 4 up to 256: AUTO pads it here instead of falling through to
    // the VALU kernel): E and dE rows are DG floats apart, columns DG .. D - 1 are read as zeros (out-of-range buffer
    // offsets) and never stored -- zero columns change neither a dot product nor a norm.
    const int DG = p.D;
    const unsigned ROWBG = (unsigned)DG * 4u;
    _Float16* const CHh = reinterpret_cast<_Float16*>(smem_f);
    _Float16* const CHl = CHh + NC * PH;
    _Float16* const ETh = CHl + NC * PH;
    _Float16* const ETl = ETh + TR * PH;
    float* const GCS = reinterpret_cast<float*>(ETh);          // [64][P] fp32 view (finalize only)
    float* const AT = reinterpret_cast<float*>(ETl + TR * PH); // U region as S / staging: [64][68] fp32
    _Float16* const Gh = reinterpret_cast<_Float16*>(AT);      // U region as G images
    _Float16* const Gl = Gh + TR * GP;
    float* const RS = AT + (2 * TR * GP * 2) / 4;
    float* const CST = RS + TR * 8;
    float* const RED = CST + NC * 4;
    static_assert(2 * TR * GP * 2 >= TR * APITCH * 4, "U region must hold the fp32 S tile");
    static_assert(2 * TR * PH * 2 >= NC * P * 4, "ET images must hold the fp32 gC staging");

    const int N = p.N, M = p.M, NM = N * M;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int d4 = 4 * lane;          // whole-wave row passes: this lane's 4 consecutive columns
    const bool dact = d4 < D;
    const int sub = lane >> 4;        // tile staging: 16 lanes per row, 4 rows per wave-instruction
    const int l16 = lane & 15;
    const int kh = wid >> 2;          // gradient contractions: 32-row (or 32-centroid) half
    const int sl = wid & 3;           //                        64-column slice of d
    const bool slice_on = 64 * sl < D;

    const int spt = wsl.spt;          // speakers per tile
    const int ntiles = wsl.ntiles;
    const unsigned ws_bytes = (unsigned)(wsl.stride * sizeof(float));
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(p.ws + (size_t)wg * wsl.stride, ws_bytes);
    const unsigned offA = (unsigned)(wsl.stash_a * 4), offR = (unsigned)(wsl.stash_rs * 4);
    const unsigned offKP = (unsigned)(wsl.dump * 4), offSM = (unsigned)(wsl.sums * 4);

    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, eps_cos = p.eps_cos, log_eps = p.log_eps;
    const float fM = (float)M, inv_m = 1.0f / (float)M, inv_m1 = 1.0f / (float)(M - 1);
    const bool contrast = p.variant == 1;
    const bool want_grad = p.dE != nullptr;

    const unsigned vrow = dact ? (unsigned)d4 * 4u : OOB;                   // one row per wave (workspace rows: D columns)
    const unsigned vrowg = d4 < DG ? (unsigned)d4 * 4u : OOB;               // ... of E: DG columns
    const unsigned vtile = (unsigned)((8 * wid + sub) * DG + 4 * l16) * 4u; // 4 rows per wave, + g*4 rows

    GE2E_PROF_DECL(10)
    bool have_sums = false;   // speaker sums of the current batch already sit in the workspace
    const int wid_outer = wid, m_outer = M, n_outer = N, spt_outer = spt, ntiles_outer = ntiles, tid_outer = tid;
    for (int bi = wg; bi < p.B; bi += nwg) {
        // wave- and shape-derived scalars are re-derived per batch from opaque copies: as loop invariants hipcc precomputes
        // ~130 of them in front of the loop, spills them to VGPR lanes and reads them back one v_readlane at a time
        int wid_o = wid_outer, m_o = m_outer, n_o = n_outer, spt_o = spt_outer, nt_o = ntiles_outer, tid_o = tid_outer;
        asm volatile("" : "+s"(wid_o), "+s"(m_o), "+s"(n_o), "+s"(spt_o), "+s"(nt_o), "+v"(tid_o));
        const int wid = wid_o, M = m_o, N = n_o, NM = N * M, spt = spt_o, ntiles = nt_o, tid = tid_o;
        const int kh = wid >> 2, sl = wid & 3;
        const bool slice_on = 64 * sl < D;
        const __amdgpu_buffer_rsrc_t rsE = make_rsrc(p.E + (size_t)bi * NM * DG, (unsigned)NM * ROWBG);
        const __amdgpu_buffer_rsrc_t rsE2s = rsE;
        const __amdgpu_buffer_rsrc_t rsE3 = rsE;
        const __amdgpu_buffer_rsrc_t rsG = make_rsrc(want_grad ? p.dE + (size_t)bi * NM * DG : nullptr,
                                                      want_grad ? (unsigned)NM * ROWBG : 0u);

        // ================= sweep 1: speaker sums -> unit centroid images =====================
        // Each wave owns a contiguous eighth of the speakers.  The first batch of this workgroup
        // streams their rows through a 16-row ring; for every later batch the sums are already in the
        // workspace: the previous batch's sweep 3 streamed these rows underneath its own compute
        // (sweep 1 alone runs at the HBM rate and used to be 16 % of the kernel with nothing to overlap).
        const int per_w = (N + NWAVE - 1) / NWAVE;
        const int jb = min(wid * per_w, N), je = min(jb + per_w, N);
        const int nr_w = (je - jb) * M;                      // rows this wave sums
        auto finish_speaker = [&](int j, const float4& s) {
            const float4 c = make_float4(s.x / fM, s.y / fM, s.z / fM, s.w / fM);
            const float sq = wave_sum(dot4(c, c));
            const float ss = wave_sum(dot4(s, s));
            float rn, kap;
            unit_stats(sq, eps_cos, rn, kap);
            if (dact) put_split4(CHh, CHl, j * PH + d4, scale4(c, rn * kSplitScale));
            if (lane == 0)  // 1/max(|c|,eps), kappa, |s_j| scale (s_j = c-hat_j * that), |s_j|^2
                *reinterpret_cast<float4*>(CST + j * 4) = make_float4(rn, kap, fM / rn, ss);
        };
        if (!have_sums) {
            const int nr = nr_w;
            const unsigned base = (unsigned)(jb * M) * ROWBG;
            constexpr int RING = 16;
            float4 ring[RING];
#pragma unroll
            for (int u = 0; u < RING; ++u)
                ring[u] = bload4<GE2E_AUX_E1>(rsE, vrowg, base + (unsigned)min(u, max(nr - 1, 0)) * ROWBG);
            float4 s = zero4();
            int cnt = 0, j = jb;
            for (int rb = 0; rb < nr; rb += RING) {
#pragma unroll
                for (int u = 0; u < RING; ++u) {
                    const int row = rb + u;
                    if (row < nr) {
                        s.x += ring[u].x; s.y += ring[u].y; s.z += ring[u].z; s.w += ring[u].w;
                        if (++cnt == M) { finish_speaker(j, s); s = zero4(); cnt = 0; ++j; }
                    }
                    ring[u] = bload4<GE2E_AUX_E1>(rsE, vrowg, base + (unsigned)min(row + RING, max(nr - 1, 0)) * ROWBG);
                }
            }
        } else {
            float4 sums[8];                                   // per_w <= 8 because N <= 64
#pragma unroll
            for (int u = 0; u < 8; ++u) sums[u] = bload4(rsW, vrow, offSM + (unsigned)min(jb + u, N - 1) * ROWB);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (jb + u < je) finish_speaker(jb + u, sums[u]);
        }
        for (int jz = N + wid; jz < NC; jz += NWAVE) {     // unused centroid slots stay zero
            if (dact) {
                *reinterpret_cast<h4*>(CHh + jz * PH + d4) = h4{0, 0, 0, 0};
                *reinterpret_cast<h4*>(CHl + jz * PH + d4) = h4{0, 0, 0, 0};
            }
            if (lane == 0) *reinterpret_cast<float4*>(CST + jz * 4) = zero4();
        }
        __syncthreads();
        GE2E_PROF(0);

        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        f32x16 gc[2];  // dL/d c-hat accumulator: centroids 32 kh.., columns 64 sl + 32 b..
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) gc[b][i] = 0.f;

        // tile rows in registers: v[g][c] = row 8 wid + 4 g + sub, columns 64 c + 4 l16 .. +3;
        // rows past the tile's last speaker get the out-of-range lane offset and read 0.
        float4 v[2][NCH];
#ifndef GE2E_AUX_E2
#define GE2E_AUX_E2 0   // sweep-2 read of E
#endif
#define GE2E_LOAD_ROWS(T) GE2E_LOAD_ROWS_AUX(T, GE2E_AUX_E2)
#define GE2E_LOAD_ROWS_AUX(T, AUX)                                                          \
    do {                                                                                    \
        const int j0_ = (T) * spt;                                                          \
        const int nrows_ = min(spt, N - j0_) * M;                                           \
        const unsigned tb_ = (unsigned)(j0_ * M) * ROWBG;                                   \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) {                                     \
            const unsigned vo_ = (8 * wid + 4 * g + sub < nrows_) ? vtile : OOB;            \
            _Pragma("unroll") for (int c = 0; c < NCH; ++c)                                 \
                v[g][c] = bload4<AUX>(rsE2s, 4 * l16 + 64 * c < DG ? vo_ : OOB, tb_ + (unsigned)(4 * g) * ROWBG + 256u * c); \
        }                                                                                   \
    } while (0)

        // ================= sweep 2: similarity rows, loss, dL/dS, gC ========================
        GE2E_LOAD_ROWS(0);
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;

            // -- (a) normalise the prefetched rows -> ET images; |e|^2, 1/|e| -> RS ---------------
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int rl = 8 * wid + 4 * g + sub;
                const bool rv = rl < nrows;
                float ee = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) ee += dot4(v[g][c], v[g][c]);
                ee = row16_sum(ee);
                float rne, ke;
                unit_stats_fast(ee, eps_cos, rne, ke);
                if (!rv) rne = 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c)
                    put_split4(ETh, ETl, rl * PH + 64 * c + 4 * l16, scale4(v[g][c], rne * kSplitScale));
                if (l16 == 0) {
                    const int j = rv ? j0 + (int)(((float)rl + 0.5f) * inv_m) : -1;
                    *reinterpret_cast<float4*>(RS + rl * 8) = make_float4(rne, ke, ee, __int_as_float(j));
                }
            }
            __syncthreads();
            // prefetch the next tile (the last iteration re-requests its own tile: no branch)
            GE2E_LOAD_ROWS(min(t + 1, ntiles - 1));
            GE2E_PROF(1);

            // -- (b) X[k][r] = sum_d CH[k][d] ET[r][d]; wave (a,b) of waves 0-3 owns k-half a, r-half b --
            if (wid < 4) {
                const int a = wid >> 1, b = wid & 1;
                const f32x16 acc = gemm_nt_32x32<D>(CHh, CHl, PH, 32 * a, ETh, ETl, PH, 32 * b, lane);
                // C layout: column (r) = lane&31, row (k) = (reg&3) + 8 (reg>>2) + 4 h
                float* Sp = AT + (32 * b + l31) * APITCH + 32 * a + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(Sp + 8 * g) =
                        make_float4(acc[4 * g] * kSplitInv2, acc[4 * g + 1] * kSplitInv2,
                                    acc[4 * g + 2] * kSplitInv2, acc[4 * g + 3] * kSplitInv2);
            }
            __syncthreads();
            GE2E_PROF(2);

            // -- (c) per row: leave-one-out stats, S = w (cos + eps) + b, loss, G = dL/dS ---------
            // 8 lanes per row, 8 centroids per lane: row reductions are three DPP steps.
            {
                const int rl = 8 * wid + (lane >> 3);
                const int qk = lane & 7;
                const float4 rs0 = *reinterpret_cast<const float4*>(RS + rl * 8);  // rne ke ee j
                const float rne = rs0.x, ke = rs0.y, ee = rs0.z;
                const int j = __float_as_int(rs0.w);
                const bool rv = j >= 0;
                const int jc = rv ? j : 0;
                const float4 cs = *reinterpret_cast<const float4*>(CST + jc * 4);  // rn kap |s| |s|^2
                const float xo = AT[rl * APITCH + jc];          // c-hat_j . e-hat_r
                const float rne1 = rv ? rne : 1.0f;
                const float es = xo * cs.z / rne1;               // e . s_j
                const float eu = (es - ee) * inv_m1;
                const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
                float rnu, ku;
                unit_stats_fast(uu, eps_cos, rnu, ku);
                const float cosd = eu * rne * rnu;               // cos(e, leave-one-out centroid)
                float c0[8];
                {
                    const float4 t0 = *reinterpret_cast<const float4*>(AT + rl * APITCH + 8 * qk);
                    const float4 t1 = *reinterpret_cast<const float4*>(AT + rl * APITCH + 8 * qk + 4);
                    c0[0] = t0.x; c0[1] = t0.y; c0[2] = t0.z; c0[3] = t0.w;
                    c0[4] = t1.x; c0[5] = t1.y; c0[6] = t1.z; c0[7] = t1.w;
                }
                const int jrel = j - 8 * qk;  // own-speaker column relative to this lane's 8
#pragma unroll
                for (int i = 0; i < 8; ++i) if (i == jrel) c0[i] = cosd;
                const float sjj = w * (cosd + eps) + bias;
                float sv[8], g[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) sv[i] = (8 * qk + i < N) ? w * (c0[i] + eps) + bias : -INFINITY;
                float per;
                if (!contrast) {
                    float mx = sv[0];
#pragma unroll
                    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, sv[i]);
                    mx = fmaxf(oct_max(mx), log_eps);
                    float zoff = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        g[i] = __expf(sv[i] - mx);  // exp(-inf) = 0 for padded centroids
                        if (i != jrel) zoff += g[i];
                    }
                    zoff = oct_sum(zoff) + __expf(log_eps - mx);
                    const float z = zoff + __expf(sjj - mx);
                    per = (mx - sjj) + __logf(z);
                    const float rz = 1.0f / z;
#pragma unroll
                    for (int i = 0; i < 8; ++i) g[i] = (i == jrel) ? -zoff * rz : g[i] * rz;  // 1 - p_jj = z_off / z
                } else {
                    float best = -INFINITY; int besti = 0x7fffffff;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (i != jrel && sv[i] > best) { best = sv[i]; besti = 8 * qk + i; }
                    oct_argmax(best, besti);
                    const float pos = 1.0f / (1.0f + __expf(-sjj));
                    const float neg = (N > 1) ? 1.0f / (1.0f + __expf(-best)) : 0.0f;
                    per = 1.0f - pos + neg;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        g[i] = (i == jrel) ? -pos * (1.0f - pos) : ((8 * qk + i == besti) ? neg * (1.0f - neg) : 0.f);
                }
                float coef = 0.f, ad = 0.f;
                const float rho_o = rnu * inv_m1;
                const float own_o = rv ? rho_o * (rne1 + ku * cosd * rho_o) * cs.z / rne1 : 0.f;   // o / (dL/dS on the own column)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (!rv || 8 * qk + i >= N) g[i] = 0.f;
                    dw_acc += g[i] * (c0[i] + eps);
                    db_acc += g[i];
                    coef += g[i] * c0[i];           // (dL/d e-hat) . e-hat / w, own-speaker term included
                    // the own-speaker column carries o = c2 |s_j| / (ra w): sweep 3's contraction adds the c2 s_j term of dE by
                    // itself, and what o adds to gC_j is, pushed through the centroid norm, exactly the leave-one-out speaker
                    // row sum_i c3_i e-hat_i minus kap_j (sum_i c3_i xo_i) c-hat_j (ge2e_team.hip, S): no second pass over the
                    // e-hat images for KJP_j
                    if (i == jrel) { ad = g[i]; g[i] *= own_o; }
                }
                coef = w * oct_sum(coef);
                ad = w * oct_sum(ad);               // dL/dcos on the own-speaker column
                // the S tile and the G images share the U region: every lane has its S values in
                // registers by now; wait for all of them before the region is rewritten
                __syncthreads();
                {
                    h4 hi0, lo0, hi1, lo1;
                    split4(make_float4(g[0] * kSplitScale, g[1] * kSplitScale, g[2] * kSplitScale, g[3] * kSplitScale), hi0, lo0);
                    split4(make_float4(g[4] * kSplitScale, g[5] * kSplitScale, g[6] * kSplitScale, g[7] * kSplitScale), hi1, lo1);
                    const h8 hh = __builtin_shufflevector(hi0, hi1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const h8 ll = __builtin_shufflevector(lo0, lo1, 0, 1, 2, 3, 4, 5, 6, 7);
                    *reinterpret_cast<h8*>(Gh + rl * GP + 8 * qk) = hh;
                    *reinterpret_cast<h8*>(Gl + rl * GP + 8 * qk) = ll;
                    // stash: [tile][hi 64x64 | lo 64x64] halfs, unpadded (16 KB per tile)
                    const unsigned va = (unsigned)(t * (TR * NC * 4) + (rl * NC + 8 * qk) * 2);
                    bstore4(rsW, va, offA, __builtin_bit_cast(float4, hh));
                    bstore4(rsW, va, offA + TR * NC * 2, __builtin_bit_cast(float4, ll));
                }
                if (rv && qk == 0) {
                    loss_acc += per;
                    if (p.per) p.per[(size_t)bi * NM + r0 + rl] = per;
                }
                {
                    // dE_r = w gE rne + c1 e-hat + c2 s_j + KJ_j   (ge2e_fused_f32.hip header), stored for
                    // sweep 3 as the ready-made coefficients of acc, of the RAW row e and of c-hat_j images
                    const float rho = rnu * inv_m1;
                    const float c2 = rho * (ad * rne1 + ad * ku * cosd * rnu * inv_m1);
                    const float c1 = (-ke * coef * rne1 - ad * rnu * inv_m1) - c2 / rne1;
                    const float alpha = ad * rnu * (1.0f + ku * cosd * rho / rne1);
                    const float beta = -ad * rnu * ku * cosd * rho;
                    if (qk == 0)     // c4': the row's share of the c-hat_j coefficient of the speaker's KJ row
                        RS[rl * 8 + 5] = rv ? inv_m1 * (beta * cs.z + cs.y * alpha * xo) : 0.f;
                    const unsigned vr = (qk == 0) ? (unsigned)((t * TR + rl) * 16) : OOB;
                    bstore4(rsW, vr, offR, make_float4(rne * (w * kSplitInv2), c1 * rne, 0.f, __int_as_float(j)));
                }
            }
            __syncthreads();
            GE2E_PROF(3);

            // -- (d0) per-speaker rows KJP_j = (sum_i c4'_i) c-hat_j, one speaker per wave; completed with dc_j / M in
            //         finalize (the e-hat part of the leave-one-out row is in gC_j already, through G's own column)
            if (want_grad && wid < nspk) {
                const int jl = wid, j = j0 + jl;
                float bsum = 0.f;
                for (int i = 0; i < M; ++i) bsum += RS[(jl * M + i) * 8 + 5];
                float4 c = zero4();
                if (dact) c = get_join4(CHh, CHl, j * PH + d4);
                const float bs = bsum * kSplitInv;
                bstore4(rsW, vrow, offKP + (unsigned)j * ROWB, make_float4(bs * c.x, bs * c.y, bs * c.z, bs * c.w));
            }
            // -- (d) gC[k][d] += sum_r G_off[r][k] ET[r][d]; wave: centroids 32 kh.., columns 64 sl.. ---
            if (slice_on && want_grad) gemm_tn_32x64(Gh, Gl, GP, 32 * kh, ETh, ETl, PH, 64 * sl, lane, gc);
            __syncthreads();
            GE2E_PROF(4);
        }

        // ---- batch scalars: fixed-order reduction over the 8 waves ---------------------------
        loss_acc = wave_sum(loss_acc);
        dw_acc = wave_sum(dw_acc);
        db_acc = wave_sum(db_acc);
        if (lane == 0) { RED[wid] = loss_acc; RED[8 + wid] = dw_acc; RED[16 + wid] = db_acc; }
        __syncthreads();
        if (tid == 0) {
            float l = 0.f, a = 0.f, c = 0.f;
#pragma unroll
            for (int i = 0; i < NWAVE; ++i) { l += RED[i]; a += RED[8 + i]; c += RED[16 + i]; }
            if (p.loss) p.loss[bi] = l;
            if (p.dw) p.dw[bi] = a;
            if (p.db) p.db[bi] = c;
        }
        if (!want_grad) { __syncthreads(); have_sums = false; continue; }

        // ---- gC -> LDS (fp32, over the ET images) -> through the centroid norm -> dc / M ----------
        if (slice_on) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int k = 32 * kh + (i & 3) + 8 * (i >> 2) + 4 * h;
                    GCS[k * P + 64 * sl + 32 * b + l31] = gc[b][i] * (w * kSplitInv2);
                }
        }
        __syncthreads();
        {
            // the complete per-speaker rows KJ_k = dc_k / M + KJP_k replace gC in place: sweep 3 does not
            // use the e-hat images, so their LDS holds all 64 rows for the epilogue (no workspace trip)
            float4 kp[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kp[u] = bload4(rsW, vrow, offKP + (unsigned)min(wid + NWAVE * u, N - 1) * ROWB);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = wid + NWAVE * u;
                if (k < N) {
                    float4 g = zero4(), c = g;
                    if (dact) {
                        g = *reinterpret_cast<const float4*>(GCS + k * P + d4);
                        c = scale4(get_join4(CHh, CHl, k * PH + d4), kSplitInv);
                    }
                    const float coef = wave_sum(dot4(g, c));
                    const float rn = CST[k * 4 + 0], kap = CST[k * 4 + 1];
                    const float f = kap * coef, sc = rn / fM;
                    if (dact)
                        *reinterpret_cast<float4*>(GCS + k * P + d4) =
                            make_float4((g.x - f * c.x) * sc + kp[u].x, (g.y - f * c.y) * sc + kp[u].y,
                                        (g.z - f * c.z) * sc + kp[u].z, (g.w - f * c.w) * sc + kp[u].w);
                }
            }
        }
        __syncthreads();
        GE2E_PROF(5);

        // ================= sweep 3: gE = G_off . CH, epilogue -> dE ===========================
        // Nothing here needs the e-hat images: the raw rows are read straight into the epilogue's own
        // layout (row 32 kh + 8 g + 4 ps + sub, columns 64 sl + 4 l16) and enter dE with the stored
        // coefficient c1 |e|^-1.  Prefetch group of a tile: those rows, the stashed G images, the
        // row scalars and this wave's speaker row KJ_j (macro, not a lambda: captured arrays go to scratch).
        float4 a4_0, a4_1, r4;
        float4 ev[4][2];
        // raw rows in the epilogue's layout: row 32 kh + 8 g + 4 h + (lane & 3), columns 64 sl + 32 b + 4 (l31 >> 2)
        const int pq = lane & 3, cq = l31 >> 2;
        const unsigned vep = slice_on ? (unsigned)((32 * kh + 4 * h + pq) * DG + 64 * sl + 4 * cq) * 4u : OOB;
#define GE2E_LOAD_TILE3(T)                                                                            \
    do {                                                                                              \
        const int t_ = (T);                                                                           \
        const unsigned ta_ = offA + (unsigned)t_ * (TR * NC * 4);                                     \
        a4_0 = bload4(rsW, (unsigned)tid * 16u, ta_);                                                 \
        a4_1 = bload4(rsW, (unsigned)tid * 16u, ta_ + TR * NC * 2);                                   \
        r4 = bload4(rsW, (unsigned)(tid & 63) * 16u, offR + (unsigned)t_ * (TR * 16));                \
    } while (0)
#define GE2E_LOAD_EROWS(T)                                                                            \
    do {                                                                                              \
        const int j0_ = (T) * spt;                                                                    \
        const int nrows_ = min(spt, N - j0_) * M;                                                     \
        const unsigned tb_ = (unsigned)(j0_ * M) * ROWBG;                                             \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                 \
            _Pragma("unroll") for (int b = 0; b < 2; ++b)                                             \
                ev[g][b] = bload4<GE2E_AUX_E3>(rsE3, (32 * kh + 8 * g + 4 * h + pq < nrows_ && 64 * sl + 32 * b + 4 * cq < DG) ? vep : OOB, \
                                               tb_ + (unsigned)(8 * g) * ROWBG + 128u * b);           \
    } while (0)

        // the next batch of this workgroup: its rows are summed per speaker underneath this sweep
        const int bnext = bi + nwg;
        const bool has_next = bnext < p.B;
        const __amdgpu_buffer_rsrc_t rsE2 = make_rsrc(p.E + (size_t)(has_next ? bnext : bi) * NM * DG,
                                                       has_next ? (unsigned)NM * ROWBG : 0u);
        const int nr2 = has_next ? nr_w : 0;
        const unsigned base2 = (unsigned)(jb * M) * ROWBG;
        float4 ring2[8];
        float4 s2 = zero4();
        int cnt2 = 0, j2 = jb, rb2 = 0;
#define GE2E_RING2_LOAD(ROW0)                                                                         \
    _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                     \
        ring2[u] = bload4<GE2E_AUX_E1>(rsE2, vrowg, base2 + (unsigned)min((ROW0) + u, max(nr2 - 1, 0)) * ROWBG)
        // 8 rows per step; a finished speaker's sum goes to the workspace (wave-uniform branch: at most
        // ceil(8 / M) + 1 stores per step, usually one or none)
#define GE2E_RING2_STEP()                                                                             \
    do {                                                                                              \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                               \
            if (rb2 + u < nr2) {                                                                      \
                s2.x += ring2[u].x; s2.y += ring2[u].y; s2.z += ring2[u].z; s2.w += ring2[u].w;       \
                if (++cnt2 == M) {                                                                    \
                    bstore4(rsW, vrow, offSM + (unsigned)j2 * ROWB, s2);                              \
                    s2 = zero4(); cnt2 = 0; ++j2;                                                     \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
        rb2 += 8;                                                                                     \
        GE2E_RING2_LOAD(rb2);                                                                         \
    } while (0)

        // prologue in the SAME relative order as inside the loop (group, ring, rows): the compiler
        // sizes each counted vmcnt wait by the path with the fewest younger operations
        GE2E_LOAD_TILE3(0);
        GE2E_RING2_LOAD(0);
        GE2E_LOAD_EROWS(0);
        for (int t = 0; t < ntiles; ++t) {
            const int j0 = t * spt;
            const int nspk = min(spt, N - j0);
            const int nrows = nspk * M;
            const int r0 = j0 * M;
            // -- (a) stage the prefetched G images, row scalars (ra c1e rc j) and speaker rows ------
            if (tid < TR) reinterpret_cast<float4*>(RS)[tid] = r4;   // RS as [64][4] in sweep 3
            // stashed images: float4 f holds 8 halfs of row f / 8 (512 float4 per image)
            *reinterpret_cast<float4*>(Gh + (tid >> 3) * GP + (tid & 7) * 8) = a4_0;
            *reinterpret_cast<float4*>(Gl + (tid >> 3) * GP + (tid & 7) * 8) = a4_1;
            __syncthreads();
            GE2E_LOAD_TILE3(min(t + 1, ntiles - 1));   // first: vmcnt retires in order, and the next (a) waits on these
            GE2E_RING2_STEP();                         // next batch's rows: 8 summed, 8 requested (HBM latency)
            GE2E_PROF(6);
            // -- (c) gE[r][d] = sum_k G_off[r][k] CH[k][d]; wave: rows 32 kh.., columns 64 sl.. ----------
            f32x16 ge[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) ge[b][i] = 0.f;
            if (slice_on) gemm_nn_32x64(Gh, Gl, GP, 32 * kh, CHh, CHl, PH, 64 * sl, lane, ge);
            GE2E_PROF(7);
            // -- (d) epilogue straight from the accumulators: a 4 x 4 in-quad transpose gives every lane
            //        four consecutive columns of ONE row, so the stores are 16 bytes wide (8 rows x 128 B
            //        per wave-instruction) with no LDS staging and no barrier after the GEMM:
            //        dE = ra acc + c1e e + KJ_j   (the c2 s_j term rides in G's own-speaker column)
            if (slice_on) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int rl = 32 * kh + 8 * g + 4 * h + pq;
                    const bool rv = rl < nrows;
                    const float4 rs = *reinterpret_cast<const float4*>(RS + rl * 4);  // ra c1e rc j
                    const int j = rv ? __float_as_int(rs.w) : j0;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float x[4] = {ge[b][4 * g], ge[b][4 * g + 1], ge[b][4 * g + 2], ge[b][4 * g + 3]};
                        quad_transpose4(x, lane);
                        const int col = 64 * sl + 32 * b + 4 * cq;
                        const float4 e = ev[g][b];
                        const float4 kj = *reinterpret_cast<const float4*>(GCS + j * P + col);
                        // pad rows get an out-of-range offset: the store is dropped, no branch
                        bstore4<GE2E_AUX_DE>(rsG, (rv && col < DG) ? (unsigned)((4 * h + pq) * DG + col) * 4u : OOB,
                                             (unsigned)(r0 + 32 * kh + 8 * g) * ROWBG,
                                make_float4(x[0] * rs.x + e.x * rs.y + kj.x, x[1] * rs.x + e.y * rs.y + kj.y,
                                            x[2] * rs.x + e.z * rs.y + kj.z, x[3] * rs.x + e.w * rs.y + kj.w));
                    }
                }
            }
            GE2E_PROF(9);
            GE2E_LOAD_EROWS(min(t + 1, ntiles - 1));   // consumed by the next epilogue; same registers
            __syncthreads();
            GE2E_PROF(8);
        }
        while (rb2 < nr2) GE2E_RING2_STEP();           // rows the tile loop did not cover (M * N/8 > 8 * ntiles)
        have_sums = has_next;
    }
    GE2E_PROF_FLUSH(10)
}
#undef GE2E_LOAD_ROWS
#undef GE2E_LOAD_ROWS_AUX
#undef GE2E_LOAD_TILE3
#undef GE2E_LOAD_EROWS
#undef GE2E_RING2_LOAD
#undef GE2E_RING2_STEP

}  // namespace fsplit
}  // namespace ge2e
