// fp32-grade contractions on the fp16 matrix cores (gfx950): every fp32 operand x is held as
//   x * 2^8 = hi + lo,   hi = fp16(x * 2^8),  lo = fp16(x * 2^8 - hi)
// and a product a.b is accumulated in fp32 as  hi.hi + hi.lo + lo.hi  (the dropped lo.lo term is
// 2^-22 relative), all three on v_mfma_f32_32x32x16_f16 into ONE accumulator scaled by 2^16.
// The fp16 products are exact in the fp32 accumulator, so the result carries ~22 mantissa bits:
// the error is of the size of fp32 rounding, at 3/16 of the fp32-MFMA cost.  The 2^8 prescale
// keeps lo out of the fp16 subnormal range for |x| >= 2^-11 (operands here are unit vectors and
// softmax probabilities, |x| <= 1).
//
// Operand images in LDS are plain row-major fp16 matrices (hi and lo separately).  A matrix that
// is contracted along its rows is read with ds_read_b128 (8 consecutive K elements per lane); the
// SAME image contracted along its columns is read with ds_read_b64_tr_b16, the hardware
// transposing read (4 K-rows x 16 columns per 16 lanes), so no second copy is kept.
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr float kSplitScale = 256.0f;             // 2^8 on every operand
constexpr float kSplitInv = 1.0f / 256.0f;
constexpr float kSplitInv2 = 1.0f / 65536.0f;     // accumulators carry 2^16

// x (already multiplied by kSplitScale) -> hi + lo
// The residual MUST be taken against the very bits that are stored as hi.  hipcc otherwise
// converts the same float twice -- v_cvt_pk_f16_f32 for the stored vector, v_cvt_f16_f32 for the
// subtraction -- and the two differ on exact ties (1 in 2^13 elements), which leaves hi + lo off
// by one fp16 ulp of hi (measured: 1e-3 absolute errors in dE on a handful of rows).  The empty
// asm makes the packed value opaque so the subtraction has to unpack it.
__device__ __forceinline__ void split4(const float4& x, h4& hi, h4& lo) {
    hi = h4{(_Float16)x.x, (_Float16)x.y, (_Float16)x.z, (_Float16)x.w};
    uint2 bits = __builtin_bit_cast(uint2, hi);
    asm volatile("" : "+v"(bits.x), "+v"(bits.y));
    hi = __builtin_bit_cast(h4, bits);
    lo = h4{(_Float16)(x.x - (float)hi[0]), (_Float16)(x.y - (float)hi[1]),
            (_Float16)(x.z - (float)hi[2]), (_Float16)(x.w - (float)hi[3])};
}
// (x * sc) -> hi + lo in ONE instruction per half: v_fma_mixlo / mixhi_f16 round the EXACT product x * sc once to fp16 (hi),
// and the residual x * sc - hi is formed fused, from the very bits stored as hi, and rounded once (lo): 8 instructions per
// float4 where scale + split4 takes 16, and the fp32 rounding of the scaled value is gone.  _u: `sc` is wave-uniform (an
// SGPR operand -- no copy into a vector register).
#define GE2E_SPLIT2_(H_, L_, X0_, X1_, SC_, C_)                                                                              \
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(H_) : "v"(X0_), C_(SC_));                                                   \
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(H_) : "v"(X1_), C_(SC_));                                                   \
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(L_) : "v"(X0_), C_(SC_), "v"(H_));       \
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(L_) : "v"(X1_), C_(SC_), "v"(H_))
__device__ __forceinline__ void split4_scaled(const float4& x, float sc, h4& hi, h4& lo) {
    uint2 h, l;
    GE2E_SPLIT2_(h.x, l.x, x.x, x.y, sc, "v");
    GE2E_SPLIT2_(h.y, l.y, x.z, x.w, sc, "v");
    hi = __builtin_bit_cast(h4, h);
    lo = __builtin_bit_cast(h4, l);
}
__device__ __forceinline__ void split4_scaled_u(const float4& x, float sc_uniform, h4& hi, h4& lo) {
    uint2 h, l;
    GE2E_SPLIT2_(h.x, l.x, x.x, x.y, sc_uniform, "s");
    GE2E_SPLIT2_(h.y, l.y, x.z, x.w, sc_uniform, "s");
    hi = __builtin_bit_cast(h4, h);
    lo = __builtin_bit_cast(h4, l);
}
// fma(float(low / high half of a packed fp16 pair), b, c) in one instruction
__device__ __forceinline__ float fma_mix_lo(unsigned hpack, float b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpack), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned hpack, float b, float c) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpack), "v"(b), "v"(c));
    return d;
}
// the two halves of split4_scaled_u as separate steps (callers interleave several vectors' chains)
__device__ __forceinline__ void split4_scaled_hi_u(const float4& x, float sc_uniform, h4& hi) {
    uint2 h;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h.x) : "v"(x.x), "s"(sc_uniform));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h.y) : "v"(x.z), "s"(sc_uniform));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h.x) : "v"(x.y), "s"(sc_uniform));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h.y) : "v"(x.w), "s"(sc_uniform));
    hi = __builtin_bit_cast(h4, h);
}
__device__ __forceinline__ void split4_scaled_lo_u(const float4& x, float sc_uniform, const h4& hi, h4& lo) {
    const uint2 h = __builtin_bit_cast(uint2, hi);
    uint2 l;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l.x) : "v"(x.x), "s"(sc_uniform), "v"(h.x));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l.y) : "v"(x.z), "s"(sc_uniform), "v"(h.y));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l.x) : "v"(x.y), "s"(sc_uniform), "v"(h.x));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l.y) : "v"(x.w), "s"(sc_uniform), "v"(h.y));
    lo = __builtin_bit_cast(h4, l);
}
__device__ __forceinline__ float4 join4(const h4& hi, const h4& lo) {  // still scaled by kSplitScale
    return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1],
                       (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]);
}

// K-contiguous fragment: 8 consecutive fp16 of one row (16-byte aligned) -> ds_read_b128
__device__ __forceinline__ h8 frag_row(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

// Transposed fragment for v_mfma_f32_32x32x16_f16 from a row-major [K][cols] image:
// this lane ends up with the 8 K-elements kb + 8 (lane>>5) + 0..7 of column cb + (lane & 31).
// ds_read_b64_tr_b16: per 16 lanes a 4 x 16 block; lane 4q+p supplies the address of block row q,
// columns 4p..4p+3; lane i receives column i.  EXEC must be all ones (callers are wave-uniform).
__device__ __forceinline__ h4 tr_read4(const _Float16* p) {
    s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(p));
    return __builtin_bit_cast(h4, r);
}
__device__ __forceinline__ h8 frag_tr(const _Float16* img, int pitch, int kb, int cb, int lane) {
    const int hh = lane >> 5, g2 = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    const _Float16* p = img + (kb + 8 * hh + q) * pitch + cb + 16 * g2 + 4 * pp;
    const h4 t0 = tr_read4(p);
    const h4 t1 = tr_read4(p + 4 * pitch);
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// acc += (ah + al) . (bh + bl) without the lo.lo term
__device__ __forceinline__ f32x16 mfma3(const h8& ah, const h8& al, const h8& bh, const h8& bl, f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    return acc;
}

// ---- the three tile contractions of the fused kernel (one wave each call) ---------------------
// C layout of every 32x32 accumulator: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).

// X[m][n] (32 x 32) = sum_{d < KD} A[am0 + m][d] * B[bn0 + n][d]; both images row-major, K contiguous.
template <int KD>
__device__ __forceinline__ f32x16 gemm_nt_32x32(const _Float16* Ah, const _Float16* Al, int pa, int am0,
                                                const _Float16* Bh, const _Float16* Bl, int pb, int bn0, int lane) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int off_a = (am0 + (lane & 31)) * pa + 8 * (lane >> 5);
    const int off_b = (bn0 + (lane & 31)) * pb + 8 * (lane >> 5);
#pragma unroll 4
    for (int s = 0; s < KD / 16; ++s)
        acc = mfma3(frag_row(Ah + off_a + 16 * s), frag_row(Al + off_a + 16 * s),
                    frag_row(Bh + off_b + 16 * s), frag_row(Bl + off_b + 16 * s), acc);
    return acc;
}

// out[a][b] (64 x 64 as 2x2 tiles) += sum_{k < 64} A[k][am0 + 32 a + m] * B[k][bn0 + 32 b + n]:
// both images row-major with K along the ROWS (A^T . B), read with the transposing load.
__device__ __forceinline__ void gemm_tn_64x64(const _Float16* Ah, const _Float16* Al, int pa, int am0,
                                              const _Float16* Bh, const _Float16* Bl, int pb, int bn0,
                                              int lane, f32x16 (&out)[2][2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = frag_tr(Ah, pa, 16 * s, am0 + 32 * t, lane);
            al[t] = frag_tr(Al, pa, 16 * s, am0 + 32 * t, lane);
            bh[t] = frag_tr(Bh, pb, 16 * s, bn0 + 32 * t, lane);
            bl[t] = frag_tr(Bl, pb, 16 * s, bn0 + 32 * t, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) out[a][b] = mfma3(ah[a], al[a], bh[b], bl[b], out[a][b]);
    }
}

// out[a][b] (64 x 64) = sum_{k < 64} A[32 a + m][k] * B[k][bn0 + 32 b + n]: A row-major with K
// contiguous (ds_read_b128), B row-major with K along the rows (transposing load).
__device__ __forceinline__ void gemm_nn_64x64(const _Float16* Ah, const _Float16* Al, int pa,
                                              const _Float16* Bh, const _Float16* Bl, int pb, int bn0,
                                              int lane, f32x16 (&out)[2][2]) {
    const int off_a = (lane & 31) * pa + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            ah[t] = frag_row(Ah + off_a + 32 * t * pa + 16 * s);
            al[t] = frag_row(Al + off_a + 32 * t * pa + 16 * s);
            bh[t] = frag_tr(Bh, pb, 16 * s, bn0 + 32 * t, lane);
            bl[t] = frag_tr(Bl, pb, 16 * s, bn0 + 32 * t, lane);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) out[a][b] = mfma3(ah[a], al[a], bh[b], bl[b], out[a][b]);
    }
}

// 8-wave variants: one 32-wide block of A against two 32-wide blocks of B (32 x 64 per wave).
// out[b] += sum_{k < 64} A[k][am0 + m] * B[k][bn0 + 32 b + n]   (both through the transposing load)
__device__ __forceinline__ void gemm_tn_32x64(const _Float16* Ah, const _Float16* Al, int pa, int am0,
                                              const _Float16* Bh, const _Float16* Bl, int pb, int bn0,
                                              int lane, f32x16 (&out)[2]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const h8 ah = frag_tr(Ah, pa, 16 * s, am0, lane), al = frag_tr(Al, pa, 16 * s, am0, lane);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            out[b] = mfma3(ah, al, frag_tr(Bh, pb, 16 * s, bn0 + 32 * b, lane),
                           frag_tr(Bl, pb, 16 * s, bn0 + 32 * b, lane), out[b]);
    }
}
// out[b] += sum_{k < 64} A[ar0 + m][k] * B[k][bn0 + 32 b + n]   (A: ds_read_b128 rows, B: transposing load)
__device__ __forceinline__ void gemm_nn_32x64(const _Float16* Ah, const _Float16* Al, int pa, int ar0,
                                              const _Float16* Bh, const _Float16* Bl, int pb, int bn0,
                                              int lane, f32x16 (&out)[2]) {
    const int off_a = (ar0 + (lane & 31)) * pa + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const h8 ah = frag_row(Ah + off_a + 16 * s), al = frag_row(Al + off_a + 16 * s);
#pragma unroll
        for (int b = 0; b < 2; ++b)
            out[b] = mfma3(ah, al, frag_tr(Bh, pb, 16 * s, bn0 + 32 * b, lane),
                           frag_tr(Bl, pb, 16 * s, bn0 + 32 * b, lane), out[b]);
    }
}

}  // namespace ge2e
