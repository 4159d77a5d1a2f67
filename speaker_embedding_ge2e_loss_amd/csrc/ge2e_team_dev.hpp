// Device helpers shared by the team kernels (ge2e_team.hip: forward + backward; ge2e_team_fwd.hip: the pipelined
// forward-only kernel): buffer-resource loads / stores, norm bookkeeping, the swizzled image layouts, MFMA wrappers.
#pragma once
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_team.hpp"
#include "ge2e_team_kernel.hpp"
#include "ge2e_fused_split_body.hpp"

namespace ge2e {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int NC = 64;        // centroid slots: member m owns 8 m .. 8 m + 7
constexpr int GP = 64;        // G image pitch (halfs): 128-byte rows, 16-byte chunks XOR-swizzled by row (g_off)
constexpr int XP = 68;        // X block pitch (floats)
constexpr int STGPAD = 32;    // centroid stage: rows 16 banks apart -> the transposing reads of a 4 x 32 block never collide
constexpr int RTMAX = 80;     // rows of a member's images
constexpr int RBMAX = RTMAX / 16;
constexpr unsigned OOB = 0x7FFFFF00u;
constexpr int AUX_L2 = 16;    // sc1: served by L2, never by this CU's L1 (hand-off reads)
constexpr int AUX_NT = 2;
#ifndef GE2E_T2_DE_AUX
#define GE2E_T2_DE_AUX 2      // cache policy of the dE stores (tools/bench_variants.py sweeps it)
#endif
// column tile i of this wave in GE / dE.  Two tiles per wave (D > 128): ADJACENT tiles, so that after the half-row
// exchange at the end of GE a lane pair-of-tiles covers whole 128-byte lines of dE (see T2_PAIR_LINES)
#define T2_DT(i) (NTI == 2 ? 2 * wid + (i) : wid + 8 * (i))
// ... of the EXCHANGE stores (published centroids in both forms; partial centroid gradients).  Every store's bytes leave the
// XCD's L2 for the fabric whatever the pressure (tools/ubench/l2_rewrite.hip: 64 rewrites of a 2-MiB-per-XCD set = 64 copies
// of WRITE_SIZE with plain or sc1 stores, 40 with nt -- the L2 merges some nt rewrites); the readers take them from L2 (sc1 loads)
#ifndef GE2E_T2_XC_AUX
#define GE2E_T2_XC_AUX 0
#endif
#ifndef GE2E_T2_GC_AUX
#define GE2E_T2_GC_AUX 0
#endif
#ifndef GE2E_T2_E_AUX
#define GE2E_T2_E_AUX 2       // ... and of the E loads
#endif

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
template <int AUX = 0>
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ h8 bload_h8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
// whole offset in the VGPR, immediate soffset (ge2e_fused_split.hip: the register-soffset store hazard)
template <int AUX = 0>
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned voff, const float4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, AUX);
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
// x / max(|x|, eps) bookkeeping without a branch: rn = 1 / max(|x|, eps), kappa = clamped / true norm (0 for a
// zero vector), nc = max(|x|, eps).  v_rsq_f32 + one Newton step instead of sqrt and two IEEE divisions.
__device__ __forceinline__ void unit_stats_bf(float sq, float eps_cos, float eps_cos2, float& rn, float& kappa, float& nc) {
    const float sqc = fmaxf(sq, eps_cos2);
    float r = __builtin_amdgcn_rsqf(sqc);
    r = r * (1.5f - 0.5f * sqc * r * r);
    rn = r;
    nc = sqc * r;
    kappa = sq >= eps_cos2 ? 1.0f : (sq > 1e-36f ? eps_cos * __builtin_amdgcn_rsqf(sq) : 0.0f);
}
__device__ __forceinline__ float rcp_nr(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}
__device__ __forceinline__ void put_split4(_Float16* hi_img, _Float16* lo_img, int off, const float4& x) {
    h4 hi, lo;
    split4(x, hi, lo);
    *reinterpret_cast<h4*>(hi_img + off) = hi;
    *reinterpret_cast<h4*>(lo_img + off) = lo;
}

// Split-fp16 products on the MFMA builtins.  (An earlier version of this file issued them as inline asm with tied
// accumulators, hand-placed wait states and operand keep-alives, after wrong 16 x 16 tiles in lanes 48..63 had looked
// like a SrcC / operand read-after-overwrite race in the matrix pipe.  The cause was elsewhere -- the SLP vectoriser's
// v_pk_mul_f32 / v_pk_fma_f32 in the fp32 epilogues, see build.py's EXTRA_FLAGS -- and with that flag the builtins pass
// the 400-launch bitwise-determinism test and every parity case; the scaffolding cost 1 % and is gone.)
__device__ __forceinline__ void mfma16(f32x4& acc, const h8& a, const h8& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfma32(f32x16& acc, const h8& a, const h8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
// acc += (ah + al) . (bh + bl) without the lo.lo term
__device__ __forceinline__ void mfma16x3(f32x4& acc, const h8& ah, const h8& al, const h8& bh, const h8& bl) {
    mfma16(acc, ah, bh); mfma16(acc, ah, bl); mfma16(acc, al, bh);
}
__device__ __forceinline__ void mfma32x3(f32x16& acc, const h8& ah, const h8& al, const h8& bh, const h8& bl) {
    mfma32(acc, ah, bh); mfma32(acc, ah, bl); mfma32(acc, al, bh);
}
__device__ __forceinline__ f32x4 acc_zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// dE leaves in WHOLE 128-byte lines.  GE's accumulator layout gives lane (r = l15, q) 16 bytes of row r in each of its
// two (adjacent) column tiles A and B: one store instruction would write 16 rows x 64 bytes, and half lines cost 29 % more
// write traffic at the fabric (rocprofv3 WRITE_SIZE 6.09 GB -> 5.30 GB per launch with whole lines, tools/
// variant_traffic.sh) and 4-8 % of the rate -- the kernel is bound by the write path.  The upper half-row of A and the
// lower half-row of B change places (DPP row_ror:8 under a bank mask, 3 VALU per dword); afterwards
//   A: row 8 * 0 + (l15 & 7), B: row 8 + (l15 & 7);  both: columns 16 dtA + 16 (l15 >> 3) + 4 q .. + 3
// so lanes {l15 & 7 = r} of one instruction write the 128 contiguous bytes of row r.
__device__ __forceinline__ void pair_lines1(float& a, float& b) {
    const int ai = __float_as_int(a), bi = __float_as_int(b);
    // lanes 0..7 of every row of 16 take A[l + 8]; lanes 8..15 keep B
    const int b2 = __builtin_amdgcn_update_dpp(bi, ai, 0x128, 0xF, 0x3, false);
    // lanes 8..15 take (old) B[l - 8]; lanes 0..7 keep A
    const int a2 = __builtin_amdgcn_update_dpp(ai, bi, 0x128, 0xF, 0xC, false);
    a = __int_as_float(a2);
    b = __int_as_float(b2);
}
#define T2_PAIR_LINES(A_, B_) \
    do { pair_lines1((A_).x, (B_).x); pair_lines1((A_).y, (B_).y); pair_lines1((A_).z, (B_).z); pair_lines1((A_).w, (B_).w); } while (0)

// ---- image layouts: rows of exactly D (64) halfs, the 16-byte chunks of a row XOR-swizzled by a function of the row --
// The round-2 images padded every row (D + 16 / 72 halfs).  That serves the ds_read_b128 row fragments, but the
// transposing reads of GC (4 rows x 32 columns per half wave) and the 8-byte e-hat reads of the epilogues were 2-way
// bank-conflicted, and so were G's row fragments: 37 % of the kernel's LDS cycles were conflict cycles.  These two
// swizzles make every read pattern of both images conflict-free (tools/lds_conflicts.py models the banks: b128 in four
// 16-lane groups, b64 / tr_b16 in two 32-lane halves, 64 banks; the G writes of S stay 2-way) and the images are 10 KB
// smaller.  An offset is in halfs; h / s (the column) must be a multiple of 4.
template <int D>
__device__ __forceinline__ int et_off(int r, int h) {
    constexpr int MASK = (D % 128 == 0) ? 15 : 7;        // D = 64, 192: stay inside an aligned group of eight chunks
    const int f = (4 * (r & 3) + ((4 - ((r >> 2) & 3)) & 3)) & MASK;
    return r * D + ((((h >> 3) ^ f) << 3) | (h & 7));
}
__device__ __forceinline__ int g_off(int r, int s) {
    const int f = (r & 3) | ((((r >> 1) ^ (r >> 2)) & 1) << 2);
    return r * GP + ((((s >> 3) ^ f) << 3) | (s & 7));
}
// transposed 32 x 32 x 16 fragment (ge2e_split_gemm.hpp: frag_tr) from a swizzled image: K-rows kb + 8 (lane >> 5) + 0..7 of
// column cb + (lane & 31)
template <int D>
__device__ __forceinline__ h8 frag_tr_et(const _Float16* img, int kb, int cb, int lane) {
    const int hh = lane >> 5, g2 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const int row = kb + 8 * hh + qq, h = cb + 16 * g2 + 4 * pp;
    const h4 t0 = tr_read4(img + et_off<D>(row, h));
    const h4 t1 = tr_read4(img + et_off<D>(row + 4, h));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ h8 frag_tr_g(const _Float16* img, int kb, int cb, int lane) {
    const int hh = lane >> 5, g2 = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
    const int row = kb + 8 * hh + qq, sl = cb + 16 * g2 + 4 * pp;
    const h4 t0 = tr_read4(img + g_off(row, sl));
    const h4 t1 = tr_read4(img + g_off(row + 4, sl));
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// one lane waits; the result travels through an LDS word that is not reused for four waits (there is a
// workgroup barrier between any two of them), so ONE barrier per wait is enough
__device__ __forceinline__ bool team_wait(const unsigned* counter, unsigned target, TeamCtl* ctl, int* sh, int& slot) {
    int* w = sh + (slot & 3);
    ++slot;
    if (threadIdx.x == 0) *w = spin_until(counter, target, ctl) ? 1 : 0;
    __syncthreads();
    return *w != 0;
}

// The call is redone by the one-workgroup-per-batch body inside this launch (team_finish said so, or the control block could
// not be trusted): workgroup `rank` of `n` takes the batches rank, rank + n, ... in workspace slice `rank` -- or, n < 0,
// every batch by itself in the slice of its block index (the end-of-grid wait ran out: what the others do is unknown; same
// results, written twice at worst).
template <int NCH>
__device__ __forceinline__ void team_redo(const Problem& p, const TeamKWs& L, const FusedWs& F, float* smem_f, int n, int rank) {
    Problem f = p;
    f.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(p.ws) + L.fb_off);
    __syncthreads();
    if (n < 0) {
        f.ws += (size_t)blockIdx.x * F.stride;
        fsplit::body<NCH>(f, F, smem_f, 0, 1);
    } else {
        if (n > L.fb_wgs) n = L.fb_wgs;              // (no more workgroups than batches)
        if (rank < n) fsplit::body<NCH>(f, F, smem_f, rank, n);
    }
}

}  // namespace

}  // namespace ge2e
