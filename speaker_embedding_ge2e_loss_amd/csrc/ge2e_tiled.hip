// GE2E_IMPL_TILED: shapes whose centroids do not fit one workgroup's LDS (N > 64 or D > 256:
// BASELINE configs 4 and 5).  Many workgroups per batch, five or six kernels on one stream, the
// three contractions on one split-fp16 MFMA tile core (below) over operand images that are split
// ONCE into fp16 hi / lo planes in the workspace:
//
//   k_prep    per speaker : sums, c-hat -> CH images + fp32 copy; per row: 1/|e|, e-hat -> EH images
//   k_sim     X = EH . CH^T                    tiles (rows x centroids), K = D                    (s3:64-70)
//   k_rows    per row     : leave-one-out cosine from X's own column, S, loss, dL/dS -> GH images (the own-speaker
//                           column carries the coefficient of s_j), row coefficients of the gradient  (s3:27, 115-127)
//   k_simrows = k_sim + k_rows in one kernel where a 256-slot tile holds a whole similarity row (128 < N <= 256): X never
//               goes to memory
//   k_gc      gC = GH^T . EH                   tiles (centroids x d), K = all N*M rows (cut into pieces where that fills
//                                              the chip; k_spk adds the partial sums)
//   k_spk     per speaker : gC through the centroid norm + a multiple of c-hat_j -> KJ rows (the leave-one-out speaker row
//                           sum_i c3_i e-hat_i is already in gC_j through GH's own column: no second pass over the rows);
//                           the batch's loss / dw / db sums
//   k_ge      gE = GH . CH, epilogue dE = ra gE + c1e e + KJ_j
//   k_reduce  loss / dw / db of forward-only calls
//
// Tile core: 128 x 128 x 64 on four waves (one LDS stage + register prefetch, two workgroups per CU), 256 x 256 x 32 on
// eight waves with two LDS stages -- register-staged (gemm_tile) or, when K is a multiple of 32, fed by
// buffer_load ... lds with one workgroup per CU walking its tiles (gemm_tile_dma / walk_tiles, GemmCfgDma).
//
// Shapes: D % 8 == 0 (rows of the fp16 planes 16-byte aligned; ragged K-steps and tiles are the cores' business), D <= 1024, N <= 1024 (row values of k_rows live in registers), any M >= 2.
// Same algebra and same split arithmetic as ge2e_fused_split.hip.  Config 5 is bound by the contractions, config
// 4 by the bytes the pipeline moves.
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_tiled.hpp"

namespace ge2e {

namespace {

constexpr int TP = 72;  // LDS tile pitch (halfs): 64 + 8

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 scale4(const float4& a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
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

// 64 x 64 fp16 tile of a row-major global image (leading dimension ld halfs) -> LDS [64][TP];
// rows >= rows_valid are zero.  256 threads, two 16-byte pieces each.
__device__ __forceinline__ void stage_tile(const _Float16* g, size_t ld, int rows_valid, _Float16* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, c8 = (idx & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < rows_valid) v = *reinterpret_cast<const uint4*>(g + (size_t)row * ld + c8);
        *reinterpret_cast<uint4*>(lds + row * TP + c8) = v;
    }
}

struct Tiles {
    _Float16 *Ah, *Al, *Bh, *Bl;
};
__device__ __forceinline__ Tiles carve_tiles(_Float16* base) {
    return Tiles{base, base + 64 * TP, base + 2 * 64 * TP, base + 3 * 64 * TP};
}


// ---- tile contraction TM x TN x 64 on the split-fp16 MFMA (one LDS stage + register prefetch) ----------------------
// Two shapes: <128,128> = 4 waves as 2 x 2, each 64 x 64 (74 KB of LDS -> two workgroups per CU), and <256,256> =
// 8 waves as 2 x 4, each 128 x 64 (147 KB, one workgroup per CU).  A 64 x 64 wave tile re-reads 682 B of fragments
// per MFMA (8 waves ask ~170 B/clk of a 128 B/clk LDS); 128 x 64 needs 512 B and the 256 x 256 stage halves the LDS
// WRITE bytes per MFMA, which cost 3-4x a read -- the big shape is for contractions with both extents >= 256.
// The global loads of K-step s + 1 are in flight while the MFMAs of K-step s issue.  An operand is either
// "K-contiguous" (global [row][K], LDS [T][KS + 8], fragments by ds_read_b128) or "K-rows" (global [K][col],
// LDS [KS][T + 16], fragments by the transposing read); KS = 64 (small tile) or 32 (big tile: its prefetch must fit
// in registers next to 128 accumulator VGPRs).
template <int TM_, int TN_>
struct GemmCfg {
    static constexpr int TM = TM_, TN = TN_;
    static constexpr int NT = TM_ == 256 ? 512 : 256;              // threads
    static constexpr int WN = TM_ == 256 ? 4 : 2;                  // waves along N (2 along M)
    static constexpr int A2 = TM_ / 64, B2 = TN_ / (32 * WN);      // 32-row / 32-column blocks per wave
    static constexpr int KS = TM_ == 256 ? 32 : 64;                // K per stage (the big tile's prefetch must fit in registers)
    static constexpr int KP = KS + 8;                              // K-contiguous tile pitch (halfs)
    static constexpr int NP = KS / 16;                             // 16-byte pieces per plane per thread (= T * KS / 8 / NT)
    static constexpr int PLANE_A = TM_ * KP > KS * (TM_ + 16) ? TM_ * KP : KS * (TM_ + 16);
    static constexpr int PLANE_B = TN_ * KP > KS * (TN_ + 16) ? TN_ * KP : KS * (TN_ + 16);
    // The big tile runs with TWO LDS stages (2 x 80 KB = the whole 160 KB, one workgroup per CU either way): the operands of
    // K-step s + 1 are written to the other stage behind the MFMAs of step s -- one barrier per K-step and no write phase in
    // which the matrix pipe idles.  The small tile keeps one stage (74 KB) so that two workgroups share a CU.
    static constexpr int STAGES = TM_ == 256 ? 2 : 1;
    static constexpr int STAGE_HALFS = 2 * (PLANE_A + PLANE_B);
    static constexpr size_t LDS_BYTES = (size_t)STAGES * STAGE_HALFS * sizeof(_Float16);
    static constexpr bool DMA = false;
};
// the big tile with its operands brought in by buffer_load ... lds (gemm_tile_dma below; K a multiple of 32)
struct GemmCfgDma : GemmCfg<256, 256> {
    static constexpr bool DMA = true;
};

struct Opnd {
    const _Float16* hi;   // plane pointers already offset to the tile's first row (K-contig) / first column (K-rows)
    const _Float16* lo;
    size_t ld;            // leading dimension of the global image (halfs)
    int valid;            // K-contig: valid tile rows; K-rows: valid tile columns
};

// T = tile extent of this operand (rows if K-contiguous, columns if K-rows)
template <class C, bool KC, int T>
__device__ __forceinline__ void gemm_fetch(const Opnd& o, int k0, int kvalid, int tid, uint4 (&rh)[C::NP], uint4 (&rl)[C::NP]) {
    static_assert(T * (C::KS / 8) == C::NP * C::NT, "pieces per plane per thread");
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        const int idx = tid + C::NT * i;
        uint4 vh = make_uint4(0, 0, 0, 0), vl = vh;
        if (KC) {   // tile [T rows][KS K]
            const int row = idx / (C::KS / 8), c8 = (idx % (C::KS / 8)) * 8;
            if (row < o.valid && c8 < kvalid) {
                vh = *reinterpret_cast<const uint4*>(o.hi + (size_t)row * o.ld + k0 + c8);
                vl = *reinterpret_cast<const uint4*>(o.lo + (size_t)row * o.ld + k0 + c8);
            }
        } else {    // tile [KS K-rows][T cols]
            const int row = idx / (T / 8), c8 = (idx % (T / 8)) * 8;
            if (row < kvalid && c8 < o.valid) {
                vh = *reinterpret_cast<const uint4*>(o.hi + (size_t)(k0 + row) * o.ld + c8);
                vl = *reinterpret_cast<const uint4*>(o.lo + (size_t)(k0 + row) * o.ld + c8);
            }
        }
        rh[i] = vh; rl[i] = vl;
    }
}
// ---- the same fetch with everything that does not change along K taken out of the K loop -----------------------------
// (rocprofv3 SQ counters at config 5, round 4: 5.0 k VALU instructions per wave and tile against 1.5 k MFMAs -- the piece
// index -> (row, column) divisions, the 64-bit address arithmetic and the bounds tests of gemm_fetch, redone every K-step.)
// A plane is a buffer resource whose base is the tile's first element; a piece's lane offset is formed once (an invalid
// row / column gets an out-of-range offset and reads zeros), a K-step only moves the SCALAR offset.
typedef unsigned int tu32x4 __attribute__((ext_vector_type(4)));
struct OpndRs {
    __amdgpu_buffer_rsrc_t hi, lo;
    unsigned kstep_bytes;      // bytes one K-step advances
};
template <class C, bool KC, int T>
__device__ __forceinline__ OpndRs gemm_rsrc(const Opnd& o, int ktotal) {
    OpndRs r;
    // the furthest byte any piece touches: K-contiguous (valid rows - 1) ld + ktotal; K-rows (ktotal - 1) ld + valid columns
    // (K-rows: up to the end of the last piece that starts inside the valid columns -- ld is a multiple of 8, the row has it)
    const size_t span = KC ? ((size_t)(o.valid - 1) * o.ld + (size_t)ktotal) * 2
                           : ((size_t)(ktotal - 1) * o.ld + (size_t)((o.valid + 7) / 8 * 8)) * 2;
    r.hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(o.hi), 0, (int)span, 0x00020000);
    r.lo = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(o.lo), 0, (int)span, 0x00020000);
    r.kstep_bytes = KC ? (unsigned)C::KS * 2u : (unsigned)((size_t)C::KS * o.ld * 2);
    return r;
}
template <class C, bool KC, int T>
__device__ __forceinline__ void gemm_piece_offsets(const Opnd& o, int tid, unsigned (&voff)[C::NP]) {
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        const int idx = tid + C::NT * i;
        if (KC) {   // tile [T rows][KS K]: a row beyond the tile's valid rows reads zeros
            const int row = idx / (C::KS / 8), c8 = (idx % (C::KS / 8)) * 8;
            voff[i] = row < o.valid ? (unsigned)(((size_t)row * o.ld + c8) * 2) : 0x7FFFFF00u;
        } else {    // tile [KS K-rows][T cols]: a column beyond the valid ones reads zeros
            const int row = idx / (T / 8), c8 = (idx % (T / 8)) * 8;
            voff[i] = c8 < o.valid ? (unsigned)(((size_t)row * o.ld + c8) * 2) : 0x7FFFFF00u;
        }
    }
}
// K-step kstep (whole: the buffer's range check covers a ragged last step -- K-contiguous pieces past ktotal lie beyond
// `span` only in the LAST valid row, so ragged K is handled by the caller falling back to gemm_fetch for that step)
template <class C>
__device__ __forceinline__ void gemm_fetch_rs(const OpndRs& r, const unsigned (&voff)[C::NP], unsigned koff,
                                              uint4 (&rh)[C::NP], uint4 (&rl)[C::NP]) {
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        rh[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r.hi, voff[i], koff, 0));
        rl[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r.lo, voff[i], koff, 0));
    }
}
template <class C, bool KC, int T>
__device__ __forceinline__ void gemm_stash_offsets(int tid, int (&soff)[C::NP]) {
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        const int idx = tid + C::NT * i;
        soff[i] = KC ? (idx / (C::KS / 8)) * C::KP + (idx % (C::KS / 8)) * 8 : (idx / (T / 8)) * (T + 16) + (idx % (T / 8)) * 8;
    }
}
template <class C>
__device__ __forceinline__ void gemm_stash_at(_Float16* th, _Float16* tl, const int (&soff)[C::NP], const uint4 (&rh)[C::NP],
                                              const uint4 (&rl)[C::NP]) {
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        *reinterpret_cast<uint4*>(th + soff[i]) = rh[i];
        *reinterpret_cast<uint4*>(tl + soff[i]) = rl[i];
    }
}
template <class C, bool KC, int T>
__device__ __forceinline__ void gemm_stash(_Float16* th, _Float16* tl, int tid, const uint4 (&rh)[C::NP], const uint4 (&rl)[C::NP]) {
#pragma unroll
    for (int i = 0; i < C::NP; ++i) {
        const int idx = tid + C::NT * i;
        const int off = KC ? (idx / (C::KS / 8)) * C::KP + (idx % (C::KS / 8)) * 8
                           : (idx / (T / 8)) * (T + 16) + (idx % (T / 8)) * 8;
        *reinterpret_cast<uint4*>(th + off) = rh[i];
        *reinterpret_cast<uint4*>(tl + off) = rl[i];
    }
}
template <class C, bool KC, int T>
__device__ __forceinline__ h8 gemm_frag(const _Float16* t, int blk0, int s, int lane) {
    if (KC) return frag_row(t + (blk0 + (lane & 31)) * C::KP + 8 * (lane >> 5) + 16 * s);
    return frag_tr(t, T + 16, 16 * s, blk0, lane);
}

// ---- the 256 x 256 tile with the operands brought into LDS by the memory pipe itself (buffer_load ... lds) ---------------
// Round 4, SQ counters of the register-staged loop at config 5: the matrix pipe busy a third of the time, LDS cycles about
// equal to MFMA cycles (every K-step each wave spends 16 ds_write_b128 = 208 cycles on the store path and the transposing
// reads of the [KS][T + 16] image collide two-way), 64 VGPRs of prefetch next to 128 accumulators.  Here a K-step's four
// planes (A hi/lo, B hi/lo: 4 x 16 KB) arrive as 64 one-KiB pieces, eight per wave, with no register in between; a piece's
// LDS bytes are lane-linear (that is what the instruction does), so the images are unpadded and the bank spread comes
// from which GLOBAL 16 bytes a lane fetches (the read side applies the same exclusive-or):
//   K-rows plane  [32 K-rows][256 cols]: 512-byte rows; 64-byte column chunk c of row r sits at chunk position c ^ (r & 3)
//                 and the two 32-byte halves of a chunk are exchanged in K-rows 8-15 and 24-31 -- a half-wave of the 16-lane
//                 transposing read covers K-rows {q, 8 + q}, q = 0..3: eight different 32-byte slots of a bank row;
//   K-contig plane [256 rows][32 K]:     64-byte rows; 16-byte piece p of row r sits at piece position p ^ kc_swizzle(r % 16)
//                 = p ^ (bit 2 of r, bit 1 of r).  ds_read_b128 does not take its lanes sixteen consecutive ones at a time:
//                 with the first form's (bit 3, bit 2) the 16-row fragments of the 16x16x32 MFMA (lane = row i, K group kg:
//                 piece kg ^ f(i)) collided two-way although sixteen consecutive lanes hit sixteen different slots
//                 (SQ_LDS_BANK_CONFLICT = half of k_sim's LDS cycles); tools/ubench/lds_b128_banks.hip times all 256
//                 linear choices of f -- this one is among the conflict-free ones, and the counter is 0 again.
// Needs ktotal % 32 == 0 (a K-contiguous piece past the end of K would read the next row); other shapes keep the loop above.
constexpr int DMA_PL = 32 * 256;            // halfs per plane
constexpr int DMA_ST = 4 * DMA_PL;          // halfs per stage (64 KB)
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef float acc4 __attribute__((ext_vector_type(4)));   // a 16 x 16 accumulator block

// K-contiguous plane: the piece position of row r's piece p is p ^ kc_swizzle(r % 16)
__device__ __forceinline__ int kc_swizzle(int r) { return (r >> 1) & 3; }
// per-lane byte offsets of this wave's two pieces (j = wave, wave + 8) of one operand's planes
template <bool KC>
__device__ __forceinline__ void dma_piece_offsets(const Opnd& o, int lane, int wid, unsigned (&vo)[2]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = wid + 8 * jj;
        if (KC) {   // piece = 16 rows x 64 B; lane -> row 16 j + lane / 4, LDS piece position lane & 3
            const int row = 16 * j + (lane >> 2), pc = (lane & 3) ^ kc_swizzle(lane >> 2);
            vo[jj] = row < o.valid ? (unsigned)(((size_t)row * o.ld + 8 * pc) * 2) : 0x7FFFFF00u;
        } else {    // piece = 2 K-rows x 512 B; lane -> row 2 j + lane / 32, LDS chunk position (lane & 31) / 4
            const int row = 2 * j + (lane >> 5), ch = ((lane & 31) >> 2) ^ (row & 3);
            const int c8 = 32 * ch + 8 * ((lane & 3) ^ ((row >> 2) & 2));
            vo[jj] = c8 < o.valid ? (unsigned)(((size_t)row * o.ld + c8) * 2) : 0x7FFFFF00u;
        }
    }
}
// The pieces are issued from inline asm: hipcc counts a `buffer_load ... lds` it knows about as a store to every LDS
// address and puts s_waitcnt vmcnt(0) in front of the next ds_read -- the fetch of K-step s + 1 would be drained before
// the first fragment of step s is read (seen in the ISA of the builtin form).  M0 (the piece's LDS byte address) is written
// in the statement that uses it; the wait is ours (dma_wait), in front of the barrier that hands the stage over.
struct DmaRs { tu32x4 hi, lo; unsigned kstep_bytes; };
template <class C, bool KC, int T>
__device__ __forceinline__ DmaRs dma_rsrc(const Opnd& o, int ktotal) {
    const size_t span = KC ? ((size_t)(o.valid - 1) * o.ld + (size_t)ktotal) * 2
                           : ((size_t)(ktotal - 1) * o.ld + (size_t)((o.valid + 7) / 8 * 8)) * 2;
    const unsigned long long ah = (unsigned long long)o.hi, al = (unsigned long long)o.lo;
    DmaRs r;
    r.hi = tu32x4{(unsigned)ah, (unsigned)(ah >> 32) & 0xFFFFu, (unsigned)span, 0x00020000u};
    r.lo = tu32x4{(unsigned)al, (unsigned)(al >> 32) & 0xFFFFu, (unsigned)span, 0x00020000u};
    r.kstep_bytes = KC ? (unsigned)C::KS * 2u : (unsigned)((size_t)C::KS * o.ld * 2);
    return r;
}
__device__ __forceinline__ void dma_piece(const tu32x4& rs, unsigned lds_byte, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dma_issue(const DmaRs& rA, const DmaRs& rB, const unsigned (&voA)[2], const unsigned (&voB)[2],
                                          unsigned kA, unsigned kB, unsigned stage_byte, int wid) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const unsigned d = stage_byte + (unsigned)(wid + 8 * jj) * 1024u;
        dma_piece(rA.hi, d, voA[jj], kA);
        dma_piece(rA.lo, d + 2u * DMA_PL, voA[jj], kA);
        dma_piece(rB.hi, d + 4u * DMA_PL, voB[jj], kB);
        dma_piece(rB.lo, d + 6u * DMA_PL, voB[jj], kB);
    }
}
// all but the n most recent memory instructions of this wave have completed (they complete in issue order)
__device__ __forceinline__ void dma_wait_but(int n) {
    if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// fragment of 16-row / 16-column block `blk` (0..15 inside the tile) of a plane for v_mfma_f32_16x16x32_f16: lane
// (i = lane % 16, kg = lane / 16) gets row / column i of the block, K = 8 kg .. 8 kg + 7 of the K-step
template <bool KC>
__device__ __forceinline__ h8 dma_frag(const _Float16* pl, int blk, int lane) {
    const int i = lane & 15, kg = lane >> 4;
    if (KC) return frag_row(pl + (16 * blk + i) * 32 + 8 * (kg ^ kc_swizzle(i)));
    const int q = i >> 2, pp = i & 3;     // the transposing read: lane 4 q + pp names K-row q, columns 4 pp .. 4 pp + 3
    const _Float16* p = pl + (8 * kg + q) * 256 + 32 * ((blk >> 1) ^ q) + 16 * ((blk ^ kg) & 1) + 4 * pp;
    const h4 t0 = tr_read4(p);
    const h4 t1 = tr_read4(p + 4 * 256);
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
#ifdef GE2E_PROFILE
#define DMA_STAMP(i)                                                          \
    do {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();         \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
        if (t_first) t_first[i] += now_ - last_;                              \
        last_ = now_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                    \
    } while (0)
#else
#define DMA_STAMP(i)
#endif
// A contraction in two calls, so that a workgroup that walks several tiles can ask for the NEXT tile's first two stages
// before it writes the current tile out (dma_begin ... epilogue ... dma_run): the ~5 k cycles in which a fresh workgroup
// waits for its first stage, and the launch of a workgroup per tile, go away.
struct DmaJob {
    DmaRs rA, rB;
    unsigned voA[2], voB[2];
    int nk;
};
template <class C, bool AKC, bool BKC>
__device__ __forceinline__ void dma_begin(DmaJob& j, const Opnd& A, const Opnd& B, int ktotal, _Float16* sm, int tid) {
    static_assert(C::TM == 256 && C::TN == 256 && C::KS == 32 && (size_t)2 * DMA_ST * 2 <= C::LDS_BYTES, "big tile only");
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    j.rA = dma_rsrc<C, AKC, C::TM>(A, ktotal);
    j.rB = dma_rsrc<C, BKC, C::TN>(B, ktotal);
    dma_piece_offsets<AKC>(A, lane, wid, j.voA);
    dma_piece_offsets<BKC>(B, lane, wid, j.voB);
    j.nk = ktotal / 32;
    const unsigned sm_byte = (unsigned)(unsigned long long)(lds_void_p)sm;
    dma_issue(j.rA, j.rB, j.voA, j.voB, 0u, 0u, sm_byte, wid);
    if (j.nk > 1) dma_issue(j.rA, j.rB, j.voA, j.voB, j.rA.kstep_bytes, j.rB.kstep_bytes, sm_byte + 2u * DMA_ST, wid);
}
// drain_all: memory instructions younger than the job's pieces may be outstanding (the previous tile's epilogue) -- wait
// for everything instead of "all but the second stage's eight pieces".
template <class C, bool AKC, bool BKC>
__device__ __forceinline__ void dma_run(const DmaJob& j, _Float16* sm, int tid, f32x16 (&acc)[C::A2][C::B2], bool drain_all,
                                        unsigned long long* t_first = nullptr) {
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6), wa = wid / C::WN, wb = wid % C::WN;
    const int ablk0 = wa * C::A2, bblk0 = wb * C::B2;
    const DmaRs& rA = j.rA;
    const DmaRs& rB = j.rB;
    const unsigned (&voA)[2] = j.voA;
    const unsigned (&voB)[2] = j.voB;
    const unsigned sm_byte = (unsigned)(unsigned long long)(lds_void_p)sm;
    // The products run on v_mfma_f32_16x16x32_f16 (one MFMA = a whole K-step of a 16 x 16 block; tools/ubench/mfma_shapes.hip:
    // with fragments coming from LDS it sustains 16-22 % more than 32x32x16, which needs the same fragment bytes but clocks
    // ~250 MHz lower under the power cap).  A wave's 128 x 64 block = 8 A blocks x 4 B blocks, 96 MFMAs per K-step.
    // Schedule of one K-step k (fragments: A 8 x (hi, lo), B 4 x (hi, lo) = 96 VGPRs, each register read once per K-step):
    //     first half : A blocks 0-3 x B, A-major   |  behind group a: read A block 4 + a of stage k
    //     wait: own pieces of stage k + 1 landed; barrier  -- everybody has read all of stage k, stage k + 1 is whole
    //     second half: A blocks 4-7 x B, B-major   |  behind group b: the pieces of K-step k + 2 into stage k's buffer, and
    //                                                 B block b, A block b of stage k + 1 (B block b is done after group b)
    // A piece has a whole K-step (~1.3 us of MFMAs) to land, no fragment read is waited for with the matrix pipe idle, one
    // barrier per K-step.  The sched_barriers pin the order (left alone hipcc moves the barrier up to the first MFMA).
    constexpr int A16 = 2 * C::A2, B16 = 2 * C::B2;
    static_assert(A16 == 8 && B16 == 4, "schedule written for the 128 x 64 wave block");
    const int a16 = 2 * ablk0, b16 = 2 * bblk0;
    const int nk = j.nk;
    unsigned kA = nk > 1 ? rA.kstep_bytes : 0u, kB = nk > 1 ? rB.kstep_bytes : 0u;
    if (nk > 1 && !drain_all) dma_wait_but(8);       // the eight pieces of step 0 (pieces land in issue order)
    else dma_wait();
    __syncthreads();
#ifdef GE2E_PROFILE
    unsigned long long last_ = 0;
    if (t_first) {   // diagnostic build: when the first stage had landed; [1..4]: the K-step's four parts, summed
        __builtin_amdgcn_sched_barrier(0);
        t_first[0] = last_ = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    acc4 c16[A16][B16];
#pragma unroll
    for (int a = 0; a < A16; ++a)
#pragma unroll
        for (int b = 0; b < B16; ++b) c16[a][b] = acc4{0.f, 0.f, 0.f, 0.f};
    h8 fa[A16][2], fb[B16][2];
    auto read_a = [&](const _Float16* st, int a) {
        fa[a][0] = dma_frag<AKC>(st, a16 + a, lane);
        fa[a][1] = dma_frag<AKC>(st + DMA_PL, a16 + a, lane);
    };
    auto read_b = [&](const _Float16* st, int b) {
        fb[b][0] = dma_frag<BKC>(st + 2 * DMA_PL, b16 + b, lane);
        fb[b][1] = dma_frag<BKC>(st + 3 * DMA_PL, b16 + b, lane);
    };
    auto mfma = [&](int a, int b) {
        c16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a][0], fb[b][0], c16[a][b], 0, 0, 0);
        c16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a][0], fb[b][1], c16[a][b], 0, 0, 0);
        c16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a][1], fb[b][0], c16[a][b], 0, 0, 0);
    };
#pragma unroll
    for (int b = 0; b < B16; ++b) read_b(sm, b);
#pragma unroll
    for (int a = 0; a < 4; ++a) read_a(sm, a);
    int cur = 0;
    for (int k = 0; k < nk; ++k, cur ^= 1) {
        const _Float16* const st = sm + cur * DMA_ST;
        const _Float16* const nx = sm + (cur ^ 1) * DMA_ST;
        const unsigned stage_byte = sm_byte + (unsigned)cur * (2u * DMA_ST);
#pragma unroll
        for (int a = 0; a < 4; ++a) {                 // A blocks 0-3 on the matrix pipe, A blocks 4-7 read behind them
            if (a == 0) __builtin_amdgcn_s_setprio(1);
            if (a == 2) __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int b = 0; b < B16; ++b) mfma(a, b);
            __builtin_amdgcn_sched_barrier(0);
            read_a(st, 4 + a);
            __builtin_amdgcn_sched_barrier(0);
        }
        DMA_STAMP(1);      // the reads of A blocks 4-7 + 48 MFMAs issued (the reads have returned)
        dma_wait();        // this wave's pieces of stage k + 1 have landed ...
        DMA_STAMP(2);
        __syncthreads();   // ... and so have everybody else's; everybody has read all of stage k
        __builtin_amdgcn_sched_barrier(0);
        DMA_STAMP(3);      // the barrier
        const bool more2 = k + 2 < nk, more1 = k + 1 < nk;
        if (more2) { kA += rA.kstep_bytes; kB += rB.kstep_bytes; }
        // Issue priority falls from barrier to barrier (3, 2 | 1, 0 over the eight MFMA groups of a K-step): of the two waves
        // that share a SIMD the one that is BEHIND wins the arbiter.  With equal priorities the older wave wins every tie,
        // finishes its K-step ~1200 cycles early and sits at the barrier while the other runs alone with every read and
        // piece it issues exposed (stamped build, views of waves 0 and 4, round 4).
#pragma unroll
        for (int b = 0; b < B16; ++b) {               // A blocks 4-7 on the matrix pipe, B-major; behind group b two pieces of
            if (b == 0) __builtin_amdgcn_s_setprio(3);    // K-step k + 2 (into stage k's buffer) and B block b, A block b of step k + 1
            if (b == 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int a = 4; a < A16; ++a) mfma(a, b);
            __builtin_amdgcn_sched_barrier(0);
            if (more2) {
                const unsigned d = stage_byte + (unsigned)(wid + 8 * (b >> 1)) * 1024u;
                if ((b & 1) == 0) {
                    dma_piece(rA.hi, d, voA[b >> 1], kA);
                    dma_piece(rA.lo, d + 2u * DMA_PL, voA[b >> 1], kA);
                } else {
                    dma_piece(rB.hi, d + 4u * DMA_PL, voB[b >> 1], kB);
                    dma_piece(rB.lo, d + 6u * DMA_PL, voB[b >> 1], kB);
                }
            }
            if (more1) { read_b(nx, b); read_a(nx, b); }
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef GE2E_PROFILE
        {   // 48 MFMAs + pieces + next step's reads issued; NOT waiting for the reads (they belong to the next step)
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();
            if (t_first) t_first[4] += now_ - last_;
            last_ = now_;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    }
    __builtin_amdgcn_s_setprio(0);
    // The epilogues (and the register-staged loop) speak the 32 x 32 accumulator layout -- lane (h = lane / 32, c = lane % 32),
    // register 4 g + j = row 8 g + 4 h + j, column c of the block.  A 16 x 16 accumulator has lane (r4 = lane / 16, c), register
    // j = row 4 r4 + j.  Two gfx950 row swaps per register pair turn the four 16 x 16 blocks X[ar][bc] of a 32 x 32 block into
    // it: v_permlane16_swap (X[ar][0], X[ar][1]) collects "h = 0 | h = 1" with lane rows (x, bc), v_permlane32_swap then
    // collects "x = 0 | x = 1" (x = g % 2) with lane rows (h, bc).  128 swaps per wave and tile, no LDS, no barrier.
#pragma unroll
    for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2)
#pragma unroll
            for (int ar = 0; ar < 2; ++ar)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float pv = c16[2 * a2 + ar][2 * b2][jj], qv = c16[2 * a2 + ar][2 * b2 + 1][jj];
                    // (element copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0, hipcc 7.2)
                    const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(pv), __float_as_uint(qv), false, false);
                    const auto s2 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
                    acc[a2][b2][4 * (2 * ar) + jj] = __uint_as_float(s2[0]);
                    acc[a2][b2][4 * (2 * ar + 1) + jj] = __uint_as_float(s2[1]);
                }
}
template <class C, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile_dma(const Opnd& A, const Opnd& B, int ktotal, _Float16* sm, int tid,
                                              f32x16 (&acc)[C::A2][C::B2], unsigned long long* t_first = nullptr) {
    DmaJob j;
    dma_begin<C, AKC, BKC>(j, A, B, ktotal, sm, tid);
    dma_run<C, AKC, BKC>(j, sm, tid, acc, false, t_first);
}

// acc[a2][b2] += A[wave rows + 32 a2 ..][K] . B[wave cols + 32 b2 ..][K]  over K = [0, ktotal)
template <class C, bool AKC, bool BKC>
__device__ __forceinline__ void gemm_tile(const Opnd& A, const Opnd& B, int ktotal, _Float16* sm, int tid,
                                          f32x16 (&acc)[C::A2][C::B2], unsigned long long* t_first = nullptr) {
    constexpr int KS = C::KS;
    const int lane = tid & 63, wid = tid >> 6, wa = wid / C::WN, wb = wid % C::WN;
    const int arow0 = wa * (32 * C::A2), bcol0 = wb * (32 * C::B2);
    if constexpr (C::DMA) {   // (a kernel holds ONE of the two loops: with both, the waits hipcc places for the register
        gemm_tile_dma<C, AKC, BKC>(A, B, ktotal, sm, tid, acc, t_first);   // prefetch of the other loop drain the pieces in flight)
        return;
    }
    uint4 pah[C::NP], pal[C::NP], pbh[C::NP], pbl[C::NP];
    gemm_fetch<C, AKC, C::TM>(A, 0, min(KS, ktotal), tid, pah, pal);
    gemm_fetch<C, BKC, C::TN>(B, 0, min(KS, ktotal), tid, pbh, pbl);
    if constexpr (C::STAGES == 2) {
        const OpndRs rA = gemm_rsrc<C, AKC, C::TM>(A, ktotal), rB = gemm_rsrc<C, BKC, C::TN>(B, ktotal);
        unsigned voA[C::NP], voB[C::NP];
        int soA[C::NP], soB[C::NP];
        gemm_piece_offsets<C, AKC, C::TM>(A, tid, voA);
        gemm_piece_offsets<C, BKC, C::TN>(B, tid, voB);
        gemm_stash_offsets<C, AKC, C::TM>(tid, soA);
        gemm_stash_offsets<C, BKC, C::TN>(tid, soB);
        gemm_stash_at<C>(sm, sm + C::PLANE_A, soA, pah, pal);
        gemm_stash_at<C>(sm + 2 * C::PLANE_A, sm + 2 * C::PLANE_A + C::PLANE_B, soB, pbh, pbl);
        __syncthreads();
        int cur = 0;
        unsigned kA = 0, kB = 0;
        for (int k0 = 0; k0 < ktotal; k0 += KS, cur ^= 1) {
            const _Float16* const Ah = sm + cur * C::STAGE_HALFS;
            const _Float16* const Al = Ah + C::PLANE_A;
            const _Float16* const Bh = Ah + 2 * C::PLANE_A;
            const _Float16* const Bl = Bh + C::PLANE_B;
            const bool more = k0 + KS < ktotal;
            if (more) {   // next K-step's operands: in flight under the MFMAs below
                kA += rA.kstep_bytes; kB += rB.kstep_bytes;
                if (k0 + 2 * KS <= ktotal) {     // a whole step: scalar K offset, no per-piece arithmetic
                    gemm_fetch_rs<C>(rA, voA, kA, pah, pal);
                    gemm_fetch_rs<C>(rB, voB, kB, pbh, pbl);
                } else {                          // the ragged last step of a K that is no multiple of KS
                    gemm_fetch<C, AKC, C::TM>(A, k0 + KS, ktotal - k0 - KS, tid, pah, pal);
                    gemm_fetch<C, BKC, C::TN>(B, k0 + KS, ktotal - k0 - KS, tid, pbh, pbl);
                }
            }
#pragma unroll
            for (int s = 0; s < KS / 16; ++s) {
                h8 bh[C::B2], bl[C::B2];
#pragma unroll
                for (int u = 0; u < C::B2; ++u) {
                    bh[u] = gemm_frag<C, BKC, C::TN>(Bh, bcol0 + 32 * u, s, lane);
                    bl[u] = gemm_frag<C, BKC, C::TN>(Bl, bcol0 + 32 * u, s, lane);
                }
#pragma unroll
                for (int a2 = 0; a2 < C::A2; ++a2) {
                    const h8 ah = gemm_frag<C, AKC, C::TM>(Ah, arow0 + 32 * a2, s, lane);
                    const h8 al = gemm_frag<C, AKC, C::TM>(Al, arow0 + 32 * a2, s, lane);
#pragma unroll
                    for (int b2 = 0; b2 < C::B2; ++b2) acc[a2][b2] = mfma3(ah, al, bh[b2], bl[b2], acc[a2][b2]);
                }
            }
            if (more) {   // ... and into the OTHER stage, which nobody has read since the barrier of the previous step
                _Float16* const nA = sm + (cur ^ 1) * C::STAGE_HALFS;
                gemm_stash_at<C>(nA, nA + C::PLANE_A, soA, pah, pal);
                gemm_stash_at<C>(nA + 2 * C::PLANE_A, nA + 2 * C::PLANE_A + C::PLANE_B, soB, pbh, pbl);
            }
            __syncthreads();
        }
        return;
    }
    _Float16* const Ah = sm;
    _Float16* const Al = sm + C::PLANE_A;
    _Float16* const Bh = sm + 2 * C::PLANE_A;
    _Float16* const Bl = Bh + C::PLANE_B;
    for (int k0 = 0; k0 < ktotal; k0 += KS) {
        gemm_stash<C, AKC, C::TM>(Ah, Al, tid, pah, pal);
        gemm_stash<C, BKC, C::TN>(Bh, Bl, tid, pbh, pbl);
        __syncthreads();
        if (k0 + KS < ktotal) {   // next K-step's operands: in flight under the MFMAs below
            gemm_fetch<C, AKC, C::TM>(A, k0 + KS, min(KS, ktotal - k0 - KS), tid, pah, pal);
            gemm_fetch<C, BKC, C::TN>(B, k0 + KS, min(KS, ktotal - k0 - KS), tid, pbh, pbl);
        }
#pragma unroll
        for (int s = 0; s < KS / 16; ++s) {
            h8 bh[C::B2], bl[C::B2];
#pragma unroll
            for (int u = 0; u < C::B2; ++u) {
                bh[u] = gemm_frag<C, BKC, C::TN>(Bh, bcol0 + 32 * u, s, lane);
                bl[u] = gemm_frag<C, BKC, C::TN>(Bl, bcol0 + 32 * u, s, lane);
            }
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2) {
                const h8 ah = gemm_frag<C, AKC, C::TM>(Ah, arow0 + 32 * a2, s, lane);
                const h8 al = gemm_frag<C, AKC, C::TM>(Al, arow0 + 32 * a2, s, lane);
#pragma unroll
                for (int b2 = 0; b2 < C::B2; ++b2) acc[a2][b2] = mfma3(ah, al, bh[b2], bl[b2], acc[a2][b2]);
            }
        }
        __syncthreads();
    }
}
template <class C>
__device__ __forceinline__ void gemm_zero(f32x16 (&acc)[C::A2][C::B2]) {
#pragma unroll
    for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a2][b2][i] = 0.f;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// k_prep: one wave per speaker.  D <= 1024: four float4 per lane cover a row.
// RM > 0: the speaker's rows (M <= RM, D <= 256 NP) stay in registers between the speaker sum and the row pass -- E is
// read from memory ONCE (the generic form reads every row twice, and with a few hundred KB of rows in flight per CU the
// second read misses the L2: FETCH_SIZE of this kernel was twice the size of E).
template <int RM, int NP>
__global__ __launch_bounds__(256) void ge2e_tiled_prep(Problem p, TiledWs L) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);   // global wave = (batch, speaker)
    const int N = p.N, M = p.M, D = p.D;
    if (gw >= p.B * N) return;
    const int bi = gw / N, j = gw - bi * N;
    const float* E = p.E + ((size_t)bi * N + j) * M * D;
    const size_t NMp = (size_t)N * M;
    _Float16* CHh = reinterpret_cast<_Float16*>(p.ws + L.ch) + (size_t)bi * 2 * N * D;
    _Float16* CHl = CHh + (size_t)N * D;
    _Float16* EHh = reinterpret_cast<_Float16*>(p.ws + L.eh) + (size_t)bi * 2 * NMp * D;
    _Float16* EHl = EHh + NMp * D;
    float* CHf = p.ws + L.chf + (size_t)bi * N * D;
    float* CST = p.ws + L.cst + ((size_t)bi * N + j) * 4;
    float* RST = p.ws + L.rst + ((size_t)bi * N + j) * M * 4;
    const float fM = (float)M;
    constexpr int NPC = RM > 0 ? NP : 4;                  // float4 chunks of a row per lane
    const int npass = RM > 0 ? NP : (D + 255) >> 8;

    float4 s[NPC];
    float4 rows[RM > 0 ? RM : 1][NPC];
#pragma unroll
    for (int c = 0; c < NPC; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RM > 0) {
#pragma unroll
        for (int i = 0; i < (RM > 0 ? RM : 1); ++i)
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
                const int d = 256 * c + 4 * lane;
                rows[i][c] = (i < M && d < D) ? *reinterpret_cast<const float4*>(E + (size_t)i * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
                s[c].x += rows[i][c].x; s[c].y += rows[i][c].y; s[c].z += rows[i][c].z; s[c].w += rows[i][c].w;
            }
    } else {
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
                const int d = 256 * c + 4 * lane;
                if (c < npass && d < D) {
                    const float4 v = *reinterpret_cast<const float4*>(E + (size_t)i * D + d);
                    s[c].x += v.x; s[c].y += v.y; s[c].z += v.z; s[c].w += v.w;
                }
            }
    }
    float sq = 0.f, ss = 0.f;   // |c|^2 with c = s / M formed first, like the reference; |s|^2 for the row stats
#pragma unroll
    for (int c = 0; c < NPC; ++c) {
        const float4 cc = make_float4(s[c].x / fM, s[c].y / fM, s[c].z / fM, s[c].w / fM);
        sq += dot4(cc, cc);
        ss += dot4(s[c], s[c]);
    }
    sq = wave_sum(sq);
    ss = wave_sum(ss);
    float rn, kap;
    unit_stats(sq, p.eps_cos, rn, kap);
    float4 chv[NPC];            // c-hat_j, this lane's columns (the rows' own-speaker cosine xo below)
#pragma unroll
    for (int c = 0; c < NPC; ++c) {
        const int d = 256 * c + 4 * lane;
        chv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < npass && d < D) {
            const float4 ch = make_float4(s[c].x / fM * rn, s[c].y / fM * rn, s[c].z / fM * rn, s[c].w / fM * rn);
            chv[c] = ch;
            *reinterpret_cast<float4*>(CHf + (size_t)j * D + d) = ch;
            h4 hi, lo;
            split4(scale4(ch, kSplitScale), hi, lo);
            *reinterpret_cast<h4*>(CHh + (size_t)j * D + d) = hi;
            *reinterpret_cast<h4*>(CHl + (size_t)j * D + d) = lo;
        }
    }
    if (lane == 0) *reinterpret_cast<float4*>(CST) = make_float4(rn, kap, fM / rn, ss);
    // rows: 1/|e| and the e-hat images
    auto row_out = [&](int i, const float4 (&v)[NPC]) {
        float red2[2] = {0.f, 0.f};      // |e|^2 and e . c-hat_j, reduced together
#pragma unroll
        for (int c = 0; c < NPC; ++c) { red2[0] += dot4(v[c], v[c]); red2[1] += dot4(v[c], chv[c]); }
        wave_sum_n<2>(red2);
        const float ee = red2[0];
        float rne, ke;
        unit_stats_fast(ee, p.eps_cos, rne, ke);
        const size_t r = (size_t)j * M + i;
#pragma unroll
        for (int c = 0; c < NPC; ++c) {
            const int d = 256 * c + 4 * lane;
            if (c < npass && d < D) {
                h4 hi, lo;
                split4(scale4(v[c], rne * kSplitScale), hi, lo);
                *reinterpret_cast<h4*>(EHh + r * D + d) = hi;
                *reinterpret_cast<h4*>(EHl + r * D + d) = lo;
            }
        }
        // rne, kappa_e, |e|^2 and xo = e-hat . c-hat_j in exact fp32 (the fused similarity + row kernel takes the own-speaker
        // cosine from here: in its tile the column sits in ONE lane of another wave)
        if (lane == 0) *reinterpret_cast<float4*>(RST + (size_t)i * 4) = make_float4(rne, ke, ee, red2[1] * rne);
    };
    if (RM > 0) {
#pragma unroll
        for (int i = 0; i < (RM > 0 ? RM : 1); ++i)
            if (i < M) row_out(i, rows[i]);
    } else {
        for (int i = 0; i < M; ++i) {
            float4 v[NPC];
#pragma unroll
            for (int c = 0; c < NPC; ++c) {
                const int d = 256 * c + 4 * lane;
                v[c] = (c < npass && d < D) ? *reinterpret_cast<const float4*>(E + (size_t)i * D + d)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            row_out(i, v);
        }
    }
}
// the instantiation for a shape: rows in registers when M <= 10 and D <= 768 (40 / 80 / 120 row registers per lane)
static void launch_prep(const Problem& p, const TiledWs& L, hipStream_t stream) {
    const dim3 grid((unsigned)((p.B * p.N + 3) / 4)), block(256);
    if (p.M <= 10 && p.D <= 256) hipLaunchKernelGGL((ge2e_tiled_prep<10, 1>), grid, block, 0, stream, p, L);
    else if (p.M <= 10 && p.D <= 512) hipLaunchKernelGGL((ge2e_tiled_prep<10, 2>), grid, block, 0, stream, p, L);
    else if (p.M <= 10 && p.D <= 768) hipLaunchKernelGGL((ge2e_tiled_prep<10, 3>), grid, block, 0, stream, p, L);
    else hipLaunchKernelGGL((ge2e_tiled_prep<0, 4>), grid, block, 0, stream, p, L);
}

// Workgroups go to the XCDs round-robin (workgroup b runs on XCD b % 8), each XCD with an L2 of its own.  Tiles are
// numbered batch-major with the tiles of one batch adjacent, so in launch order a batch's tiles -- which share an operand
// (the centroid planes in k_sim / k_ge) -- land on eight different L2s and the shared operand is fetched eight times.
// This gives the workgroups of ONE XCD a contiguous range of tile numbers instead.
__device__ __forceinline__ int xcd_major_tile(unsigned b, unsigned grid) {
    const unsigned per = grid / 8;
    return b < per * 8 ? (int)((b % 8) * per + b / 8) : (int)b;
}

// One workgroup, several tiles (the DMA configuration only; launched with min(tiles, CUs) workgroups): `decode(t, A, B, ix)`
// sets a tile's operands and indices, `epilogue(ix, acc)` writes it out.  The next tile's first two stages are requested
// before the current tile's epilogue; virtual workgroup numbers b, b + grid, ... keep a workgroup on one XCD's contiguous
// range of tiles (xcd_major_tile below; grid is a multiple of 8 or a single pass).
template <class C, bool AKC, bool BKC, class Ix, class Decode, class Epilogue>
__device__ __forceinline__ void walk_tiles(int ntiles, _Float16* sm, int tid, Decode decode, Epilogue epilogue,
                                           unsigned long long* t_first = nullptr) {
    f32x16 acc[C::A2][C::B2];
    if constexpr (!C::DMA) {    // one tile per workgroup
        Opnd A, Bo; Ix ix; int ktotal;
        decode(xcd_major_tile(blockIdx.x, gridDim.x), A, Bo, ix, ktotal);
        gemm_zero<C>(acc);
        gemm_tile<C, AKC, BKC>(A, Bo, ktotal, sm, tid, acc);
        epilogue(ix, acc);
    } else {
        int lin = blockIdx.x;
        Opnd A, Bo; Ix ix; int ktotal;
        decode(xcd_major_tile(lin, ntiles), A, Bo, ix, ktotal);
        DmaJob j;
        dma_begin<C, AKC, BKC>(j, A, Bo, ktotal, sm, tid);
        bool first = true;
        for (;;) {
            gemm_zero<C>(acc);
            dma_run<C, AKC, BKC>(j, sm, tid, acc, !first, t_first);
            first = false;
            const int nlin = lin + (int)gridDim.x;
            const Ix cur = ix;
            if (nlin < ntiles) {   // (every wave is past the last barrier of the loop: nobody reads the stages any more)
                decode(xcd_major_tile(nlin, ntiles), A, Bo, ix, ktotal);
                dma_begin<C, AKC, BKC>(j, A, Bo, ktotal, sm, tid);
            }
            epilogue(cur, acc);
            if (nlin >= ntiles) break;
            lin = nlin;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_sim: X[r][k] = sum_d EH[r][d] CH[k][d].  One workgroup per TM x TN tile (both operands K-contiguous).
template <class C>
__global__ __launch_bounds__(C::NT, 2) void ge2e_tiled_sim(Problem p, TiledWs L) {
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = p.N, D = p.D, NM = p.N * p.M;
    const int rt = (NM + C::TM - 1) / C::TM, ct = (N + C::TN - 1) / C::TN;
    const size_t NMp = (size_t)NM;
    struct Ix { int kt, rtile, bi; };
    auto decode = [&](int t, Opnd& A, Opnd& Bo, Ix& ix, int& ktotal) {
        ix.kt = t % ct; t /= ct;
        ix.rtile = t % rt;
        ix.bi = t / rt;
        A.hi = reinterpret_cast<const _Float16*>(p.ws + L.eh) + (size_t)ix.bi * 2 * NMp * D + (size_t)ix.rtile * C::TM * D;
        A.lo = A.hi + NMp * D;
        A.ld = D; A.valid = min(C::TM, NM - ix.rtile * C::TM);
        Bo.hi = reinterpret_cast<const _Float16*>(p.ws + L.ch) + (size_t)ix.bi * 2 * N * D + (size_t)ix.kt * C::TN * D;
        Bo.lo = Bo.hi + (size_t)N * D;
        Bo.ld = D; Bo.valid = min(C::TN, N - ix.kt * C::TN);
        ktotal = D;
    };
    GE2E_PROF_DECL(8)
    unsigned long long t_first[5] = {0, 0, 0, 0, 0};
    auto epilogue = [&](const Ix& ix, f32x16 (&acc)[C::A2][C::B2]) {
        GE2E_PROF_AT(0, t_first[0]);
        GE2E_PROF(1);
        float* X = p.ws + L.x + (size_t)ix.bi * NMp * L.npad;
        const int l31 = lane & 31, h = lane >> 5, wa = wid / C::WN, wb = wid % C::WN, pq = lane & 3, cq = l31 >> 2;
#pragma unroll
        for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = ix.rtile * C::TM + wa * (32 * C::A2) + 32 * a2 + 8 * g + 4 * h + pq;
#pragma unroll
                for (int b2 = 0; b2 < C::B2; ++b2) {
                    float x[4] = {acc[a2][b2][4 * g], acc[a2][b2][4 * g + 1], acc[a2][b2][4 * g + 2], acc[a2][b2][4 * g + 3]};
                    quad_transpose4(x, lane);
                    const int k = ix.kt * C::TN + wb * (32 * C::B2) + 32 * b2 + 4 * cq;
                    if (r < NM && k < L.npad)
                        *reinterpret_cast<float4*>(X + (size_t)r * L.npad + k) =
                            make_float4(x[0] * kSplitInv2, x[1] * kSplitInv2, x[2] * kSplitInv2, x[3] * kSplitInv2);
                }
            }
        GE2E_PROF(2);
    };
    walk_tiles<C, true, true, Ix>(p.B * rt * ct, gsm, tid, decode, epilogue, t_first);
#ifdef GE2E_PROFILE
    for (int i = 0; i < 4; ++i) prof_acc[4 + i] = t_first[1 + i];
#endif
    GE2E_PROF_FLUSH_AT(0, 8)
}

// ---------------------------------------------------------------------------------------------
// k_simrows (N <= 256, i.e. one 256-slot tile holds a row's whole similarity vector): k_sim and k_rows16 in ONE kernel --
// the fp32 similarity block never goes to memory (2 x N M npad 4 bytes per batch less: 1.3 GB per launch at config 4,
// where every kernel of this pipeline runs at the HBM rate) and one launch fewer.
// The contraction is taken with the SLOTS as the A operand (C layout: register = slot, lane = row), so a row's 256
// similarities are 64 registers of TWO lanes (l, l + 32) of TWO waves (wa = 0, 1): the row reductions are in-lane loops,
// one half swap and one exchange through LDS.  The dL/dS tile then goes through LDS (the GEMM's stages are free by
// then) so that the GH planes are written as whole rows.
// CONTRAST / FULL (N == 256: every slot of the tile exists) are compile-time: the three passes over the 128 accumulators of a
// lane are the kernel's vector work, and every per-element test in them is two instructions.
template <class C, bool CONTRAST, bool FULL>
__global__ __launch_bounds__(C::NT, 2) void ge2e_tiled_simrows(Problem p, TiledWs L) {
    static_assert(C::TM == 256 && C::TN == 256, "one 256 x 256 tile per workgroup");
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = p.N, M = p.M, D = p.D, NM = N * M, npad = L.npad;
    const int rt = (NM + C::TN - 1) / C::TN;
    const int t = xcd_major_tile(blockIdx.x, gridDim.x);
    const int rtile = t % rt, bi = t / rt;
    const size_t NMp = (size_t)NM;
    Opnd A, Bo;
    A.hi = reinterpret_cast<const _Float16*>(p.ws + L.ch) + (size_t)bi * 2 * N * D;
    A.lo = A.hi + (size_t)N * D;
    A.ld = D; A.valid = min(C::TM, N);
    Bo.hi = reinterpret_cast<const _Float16*>(p.ws + L.eh) + (size_t)bi * 2 * NMp * D + (size_t)rtile * C::TN * D;
    Bo.lo = Bo.hi + NMp * D;
    Bo.ld = D; Bo.valid = min(C::TN, NM - rtile * C::TN);
    f32x16 acc[C::A2][C::B2];
    gemm_zero<C>(acc);
    gemm_tile<C, true, true>(A, Bo, D, gsm, tid, acc);      // ends with a barrier: the stages are free

    // acc[a2][b2][v] = 2^16 X[slot][row], slot = 128 wa + 32 a2 + 8 (v >> 2) + 4 h + (v & 3), row = 64 wb + 32 b2 + l31
    const int l31 = lane & 31, h = lane >> 5, wa = wid / C::WN, wb = wid % C::WN;
    float* const XCH = reinterpret_cast<float*>(gsm);                 // [2 (wa)][256 rows][2] exchange of row partials
    _Float16* const TT = gsm + 4096;                                   // [256 rows][TTP] halfs: one G plane at a time
    constexpr int TTP = 256 + 4;
    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, log_eps = p.log_eps;
    const float inv_m1 = 1.0f / (float)(M - 1);
    const int soff = 128 * wa + 4 * h;                                 // slot of (a2 = 0, v = 0)

    float rne[2], ke[2], cosd[2], sjj[2], rnu[2], ku[2], xo[2], csx[2], csy[2], csz[2];
    int jl[2];          // own-speaker slot relative to soff (matches 32 a2 + 8 (v >> 2) + (v & 3) of at most one register)
    bool rv[2];
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        const int rloc = 64 * wb + 32 * b2 + l31;
        const int r = rtile * C::TN + rloc;
        rv[b2] = r < NM;
        const int rc = rv[b2] ? r : NM - 1;
        const int j = rc / M;
        const float4 rst = *reinterpret_cast<const float4*>(p.ws + L.rst + ((size_t)bi * NM + rc) * 4);   // rne ke ee xo
        const float4 cs = *reinterpret_cast<const float4*>(p.ws + L.cst + ((size_t)bi * N + j) * 4);       // rn kap |s| |s|^2
        rne[b2] = rst.x; ke[b2] = rst.y; xo[b2] = rst.w;
        csx[b2] = cs.x; csy[b2] = cs.y; csz[b2] = cs.z;
        const float ee = rst.z;
        const float es = rst.w * cs.z / rst.x;
        const float eu = (es - ee) * inv_m1;
        const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
        unit_stats_fast(uu, p.eps_cos, rnu[b2], ku[b2]);
        cosd[b2] = eu * rst.x * rnu[b2];
        sjj[b2] = w * (cosd[b2] + eps) + bias;
        jl[b2] = j - soff;
    }
    const int nl = N - soff;        // slots of this lane with (32 a2 + 8 (v >> 2) + (v & 3)) < nl exist
    const float ws = w * kSplitInv2, bs = w * eps + bias;       // S = ws acc + bs

    // ---- row maximum over the other speakers' columns (the own column enters with its leave-one-out value) ----------
    // (jl and nl are re-read through an opaque copy in every phase: compared once, hipcc keeps all 128 (exists, own-column)
    // lane masks of the three loops below in SGPRs -- 250 spilled scalars)
#define GE2E_SR_OPAQUE() int jq[2] = {jl[0], jl[1]}, nq = nl; asm volatile("" : "+v"(jq[0]), "+v"(jq[1]), "+v"(nq))
    float best[2]; int besti[2];
    float mx[2];
    {
    GE2E_SR_OPAQUE();
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        float m = -INFINITY;
        best[b2] = -INFINITY; besti[b2] = 0x7fffffff;
        if (!CONTRAST) {       // (the two variants as separate straight-line loops: no per-element control flow)
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = 32 * a2 + 8 * (v >> 2) + (v & 3);
                    const float sv = fmaf(ws, acc[a2][b2][v], bs);
                    m = ((FULL || c < nq) && c != jq[b2]) ? fmaxf(m, sv) : m;
                }
        } else {
            float bv = -INFINITY; int bi_ = 0x7fffffff;
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = 32 * a2 + 8 * (v >> 2) + (v & 3);     // ascending slot order: ties keep the lower slot
                    const float sv = fmaf(ws, acc[a2][b2][v], bs);
                    const bool better = (FULL || c < nq) && c != jq[b2] && sv > bv;
                    bv = better ? sv : bv;
                    bi_ = better ? soff + c : bi_;
                }
            best[b2] = bv; besti[b2] = bi_;
        }
        {   // the two half-rows of the wave
            auto sw = GE2E_SWAP32(__float_as_uint(m));
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        mx[b2] = m;
    }
    }
    if (CONTRAST) {
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2) {
            auto a = GE2E_SWAP32(__float_as_uint(best[b2]));
            auto b = GE2E_SWAP32((unsigned)besti[b2]);
            float v0 = __uint_as_float(a[0]); int i0 = (int)b[0];
            argmax_merge(v0, i0, __uint_as_float(a[1]), (int)b[1]);
            best[b2] = v0; besti[b2] = i0;
        }
    }
    if (h == 0) {
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2) {
            const int rloc = 64 * wb + 32 * b2 + l31;
            XCH[(wa * 256 + rloc) * 2] = CONTRAST ? best[b2] : mx[b2];
            XCH[(wa * 256 + rloc) * 2 + 1] = __int_as_float(besti[b2]);
        }
    }
    __syncthreads();
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        const int rloc = 64 * wb + 32 * b2 + l31;
        const float om = XCH[((wa ^ 1) * 256 + rloc) * 2];
        if (CONTRAST) {
            const int oi = __float_as_int(XCH[((wa ^ 1) * 256 + rloc) * 2 + 1]);
            argmax_merge(best[b2], besti[b2], om, oi);
        } else {
            mx[b2] = fmaxf(fmaxf(fmaxf(mx[b2], om), sjj[b2]), log_eps);
        }
    }
    __syncthreads();

    // ---- exponentials, row sums -> dL/dS in place of the similarities ----------------------------------------------
    float zl[2], al[2];
    {
    GE2E_SR_OPAQUE();
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        float z = 0.f, a = 0.f;
        if (!CONTRAST) {
            const float tb2 = (bs - mx[b2]) * 1.44269504088896341f, ws2 = ws * 1.44269504088896341f;   // base 2: one instruction per exponential
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = 32 * a2 + 8 * (v >> 2) + (v & 3);
                    const float x = acc[a2][b2][v];
                    float g = __builtin_amdgcn_exp2f(fmaf(ws2, x, tb2));
                    g = ((FULL || c < nq) && c != jq[b2]) ? g : 0.f;
                    z += g;
                    a = fmaf(g, x, a);
                    acc[a2][b2][v] = g;
                }
        } else {
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = 32 * a2 + 8 * (v >> 2) + (v & 3);
                    const bool hit = soff + c == besti[b2] && (FULL || c < nq) && c != jq[b2];
                    a = hit ? acc[a2][b2][v] : a;          // the raw similarity of the best other speaker
                    acc[a2][b2][v] = hit ? 1.0f : 0.f;
                }
        }
        {
            auto s0 = GE2E_SWAP32(__float_as_uint(z));
            z = __uint_as_float(s0[0]) + __uint_as_float(s0[1]);
            auto s1 = GE2E_SWAP32(__float_as_uint(a));
            a = __uint_as_float(s1[0]) + __uint_as_float(s1[1]);
        }
        zl[b2] = z; al[b2] = a;
    }
    }
    if (h == 0) {
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2) {
            const int rloc = 64 * wb + 32 * b2 + l31;
            XCH[(wa * 256 + rloc) * 2] = zl[b2];
            XCH[(wa * 256 + rloc) * 2 + 1] = al[b2];
        }
    }
    __syncthreads();
    float gs[2], og[2];     // scale of the other columns' values, value of the own column (both x 2^8)
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        const int rloc = 64 * wb + 32 * b2 + l31;
        // fixed order (wa = 0 first): both waves of a row form the same sums
        const float z0 = wa == 0 ? zl[b2] : XCH[rloc * 2], z1 = wa == 1 ? zl[b2] : XCH[(256 + rloc) * 2];
        const float a0 = wa == 0 ? al[b2] : XCH[rloc * 2 + 1], a1 = wa == 1 ? al[b2] : XCH[(256 + rloc) * 2 + 1];
        const float zp = z0 + z1, ap = (a0 + a1) * kSplitInv2;
        float per, ad0, coefsum, db_row, gsc;
        if (!CONTRAST) {
            const float zoff = zp + __expf(log_eps - mx[b2]);
            const float z = zoff + __expf(sjj[b2] - mx[b2]);
            per = (mx[b2] - sjj[b2]) + __logf(z);
            const float rz = 1.0f / z;
            ad0 = -zoff * rz;                         // dL/dS on the own-speaker column: -(1 - p_jj) = -z_off / z
            coefsum = fmaf(ap, rz, ad0 * cosd[b2]);   // sum_k dL/dS_k c0_k
            db_row = fmaf(zp, rz, ad0);
            gsc = rz;
        } else {
            const float pos = 1.0f / (1.0f + __expf(-sjj[b2]));
            const float neg = (N > 1) ? 1.0f / (1.0f + __expf(-best[b2])) : 0.0f;
            per = 1.0f - pos + neg;
            ad0 = -pos * (1.0f - pos);
            const float gn = neg * (1.0f - neg);
            coefsum = fmaf(gn, ap, ad0 * cosd[b2]);   // (a0 + a1) is the raw similarity of the best other speaker
            db_row = gn + ad0;
            gsc = gn;
        }
        if (!rv[b2]) { ad0 = 0.f; gsc = 0.f; }
        const float rho = rnu[b2] * inv_m1, t1 = ku[b2] * cosd[b2] * rho;
        // the own-speaker column carries o = c2 |s_j| / (ra w)  (k_rows16 above)
        og[b2] = rho * (rne[b2] + t1) * csz[b2] / rne[b2] * ad0 * kSplitScale;
        gs[b2] = gsc * kSplitScale;
        if (wa == 0 && h == 0 && rv[b2]) {
            const size_t gr = (size_t)bi * NM + rtile * C::TN + rloc;
            const float coef = w * coefsum, ad = w * ad0;
            const float c2 = rho * (ad * rne[b2] + ad * ku[b2] * cosd[b2] * rnu[b2] * inv_m1);
            const float c1 = (-ke[b2] * coef * rne[b2] - ad * rnu[b2] * inv_m1) - c2 / rne[b2];
            const float alpha = ad * rnu[b2] * (1.0f + ku[b2] * cosd[b2] * rho / rne[b2]);
            const float beta = -ad * rnu[b2] * ku[b2] * cosd[b2] * rho;
            float* rs = p.ws + L.rs + gr * 8;
            *reinterpret_cast<float4*>(rs) = make_float4(rne[b2] * (w * kSplitInv2), c1 * rne[b2], 0.f, 0.f);
            *reinterpret_cast<float4*>(rs + 4) = make_float4(inv_m1 * (beta * csz[b2] + csy[b2] * alpha * xo[b2]), per,
                                                              fmaf(eps, db_row, coefsum), db_row);
            if (p.per) p.per[gr] = per;
        }
    }
    if (!p.dE) return;      // forward-only calls: nobody reads the dL/dS planes (a third of this kernel's time and 2.7 MB per tile)
    __syncthreads();        // XCH has been read: the region below it is about to hold the G tile

    // ---- dL/dS -> split fp16 -> LDS tile [row][slot] -> the GH planes as whole rows, one plane at a time ------------
    _Float16* const GHh = reinterpret_cast<_Float16*>(p.ws + L.gh) + (size_t)bi * 2 * NMp * npad + (size_t)rtile * C::TN * npad;
    const int rows_here = min(C::TN, NM - rtile * C::TN);
    // The tile is formed ONCE: the hi halves go to the LDS tile, the lo halves are parked in the accumulator registers they
    // came from (two registers per four values) until the hi plane has left.  (Round 4's first form ran the select, the scale
    // and the whole split once per plane: ~1 000 of the kernel's 4 100 vector instructions per wave.)
    auto write_plane = [&](int plane) {
        __syncthreads();
        _Float16* const G = GHh + (size_t)plane * NMp * npad;
        for (int idx = tid; idx < 256 * 32; idx += C::NT) {          // 32 16-byte pieces per row
            const int row = idx >> 5, c8 = (idx & 31) * 8;
            if (row < rows_here && c8 < npad) {
                const uint2 lo8 = *reinterpret_cast<const uint2*>(TT + row * TTP + c8);
                const uint2 hi8 = *reinterpret_cast<const uint2*>(TT + row * TTP + c8 + 4);
                *reinterpret_cast<uint4*>(G + (size_t)row * npad + c8) = make_uint4(lo8.x, lo8.y, hi8.x, hi8.y);
            }
        }
        __syncthreads();
    };
    {
        GE2E_SR_OPAQUE();
        (void)nq;
#pragma unroll
        for (int b2 = 0; b2 < C::B2; ++b2) {
            const int rloc = 64 * wb + 32 * b2 + l31;
#pragma unroll
            for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
                for (int vg = 0; vg < 4; ++vg) {
                    float gv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 32 * a2 + 8 * vg + e;
                        gv[e] = c == jq[b2] ? og[b2] : acc[a2][b2][4 * vg + e] * gs[b2];
                    }
                    h4 hi, lo;
                    split4(make_float4(gv[0], gv[1], gv[2], gv[3]), hi, lo);
                    *reinterpret_cast<h4*>(TT + rloc * TTP + soff + 32 * a2 + 8 * vg) = hi;
                    const uint2 lb = __builtin_bit_cast(uint2, lo);
                    acc[a2][b2][4 * vg] = __uint_as_float(lb.x);
                    acc[a2][b2][4 * vg + 1] = __uint_as_float(lb.y);
                }
        }
    }
    write_plane(0);
#pragma unroll
    for (int b2 = 0; b2 < C::B2; ++b2) {
        const int rloc = 64 * wb + 32 * b2 + l31;
#pragma unroll
        for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
            for (int vg = 0; vg < 4; ++vg)
                *reinterpret_cast<uint2*>(TT + rloc * TTP + soff + 32 * a2 + 8 * vg) =
                    make_uint2(__float_as_uint(acc[a2][b2][4 * vg]), __float_as_uint(acc[a2][b2][4 * vg + 1]));
    }
    write_plane(1);
#undef GE2E_SR_OPAQUE
}

// ---------------------------------------------------------------------------------------------
// k_rows: one wave per row; the row of X (<= 1024 centroids) lives in 16 registers per lane (4 x 4 consecutive slots).
__global__ __launch_bounds__(256) void ge2e_tiled_rows(Problem p, TiledWs L) {
    const int lane = threadIdx.x & 63;
    const size_t gr = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // global row
    const int N = p.N, M = p.M, NM = N * M, npad = L.npad;
    if (gr >= (size_t)p.B * NM) return;
    const int bi = (int)(gr / NM), r = (int)(gr - (size_t)bi * NM), j = r / M;
    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, log_eps = p.log_eps;
    const float inv_m1 = 1.0f / (float)(M - 1);
    const float* X = p.ws + L.x + ((size_t)bi * NM + r) * npad;
    const float4 rst = *reinterpret_cast<const float4*>(p.ws + L.rst + gr * 4);      // rne ke ee
    const float4 cs = *reinterpret_cast<const float4*>(p.ws + L.cst + ((size_t)bi * N + j) * 4);  // rn kap |s| |s|^2
    const float rne = rst.x, ke = rst.y, ee = rst.z;
    const float xo = X[j];
    const float es = xo * cs.z / rne;
    const float eu = (es - ee) * inv_m1;
    const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
    float rnu, ku;
    unit_stats_fast(uu, p.eps_cos, rnu, ku);
    const float cosd = eu * rne * rnu;
    const float sjj = w * (cosd + eps) + bias;
    // lane l holds the 4 consecutive centroid slots 256 c + 4 l .. + 3 of up to four 256-slot chunks: 16-byte reads
    // of X and 8-byte writes of the two G planes (2-byte stores cost ~12x per byte)
    float c0[4][4], g[4][4];
    float mx = -INFINITY, best = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kb = 256 * c + 4 * lane;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb < npad) xv = *reinterpret_cast<const float4*>(X + kb);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kb + e;
            c0[c][e] = (k == j) ? cosd : xe[e];
            if (k < N) {
                const float sv = w * (c0[c][e] + eps) + bias;
                mx = fmaxf(mx, sv);
                if (k != j && sv > best) { best = sv; besti = k; }
            }
        }
    }
    float per;
    if (p.variant == 0) {
        mx = fmaxf(wave_max(mx), log_eps);
        float zoff = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 256 * c + 4 * lane + e;
                g[c][e] = (k < N) ? __expf(w * (c0[c][e] + eps) + bias - mx) : 0.f;
                if (k != j) zoff += g[c][e];
            }
        zoff = wave_sum(zoff) + __expf(log_eps - mx);
        const float z = zoff + __expf(sjj - mx);
        per = (mx - sjj) + __logf(z);
        const float rz = 1.0f / z;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[c][e] = (256 * c + 4 * lane + e == j) ? -zoff * rz : g[c][e] * rz;
    } else {
        wave_argmax(best, besti);
        const float pos = 1.0f / (1.0f + __expf(-sjj));
        const float neg = (N > 1) ? 1.0f / (1.0f + __expf(-best)) : 0.0f;
        per = 1.0f - pos + neg;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 256 * c + 4 * lane + e;
                g[c][e] = (k == j) ? -pos * (1.0f - pos) : ((k == besti) ? neg * (1.0f - neg) : 0.f);
            }
    }
    float dwv = 0.f, dbv = 0.f, coef = 0.f, ad = 0.f;
    const float rho_ = rnu * inv_m1, t1_ = ku * cosd * rho_;
    const float own_o = rho_ * (rne + t1_) * cs.z / rne;     // o / (dL/dS on the own column)
    _Float16* GHh = reinterpret_cast<_Float16*>(p.ws + L.gh) + (size_t)bi * 2 * NM * npad + (size_t)r * npad;
    _Float16* GHl = GHh + (size_t)NM * npad;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kb = 256 * c + 4 * lane;
        if (kb < npad) {
            float gv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                gv[e] = (k < N) ? g[c][e] : 0.f;
                dwv += gv[e] * (c0[c][e] + eps);
                dbv += gv[e];
                coef += gv[e] * c0[c][e];
                // the own-speaker column carries o = c2 |s_j| / (ra w): k_ge's contraction then adds the c2 s_j term of dE
                // by itself, and what o adds to gC_j is, pushed through the centroid norm in k_spk, exactly the
                // leave-one-out speaker row sum_i c3_i e-hat_i minus kap_j (sum_i c3_i xo_i) c-hat_j (ge2e_team.hip, S)
                if (k == j) { ad = gv[e]; gv[e] = own_o * gv[e]; }
            }
            if (p.dE) {      // (forward-only calls: nobody reads the dL/dS planes)
                h4 hi, lo;
                split4(make_float4(gv[0] * kSplitScale, gv[1] * kSplitScale, gv[2] * kSplitScale, gv[3] * kSplitScale), hi, lo);
                *reinterpret_cast<h4*>(GHh + kb) = hi;
                *reinterpret_cast<h4*>(GHl + kb) = lo;
            }
        }
    }
    dwv = wave_sum(dwv); dbv = wave_sum(dbv);
    coef = w * wave_sum(coef); ad = w * wave_sum(ad);
    if (lane == 0) {
        const float rho = rnu * inv_m1;
        const float c2 = rho * (ad * rne + ad * ku * cosd * rnu * inv_m1);
        const float c1 = (-ke * coef * rne - ad * rnu * inv_m1) - c2 / rne;
        const float alpha = ad * rnu * (1.0f + ku * cosd * rho / rne);
        const float beta = -ad * rnu * ku * cosd * rho;
        float* rs = p.ws + L.rs + gr * 8;
        // ra, c1e; then the row's share of the c-hat_j coefficient of KJ_j: c4' = (beta |s_j| + kap_j alpha xo) / (M - 1)
        *reinterpret_cast<float4*>(rs) = make_float4(rne * (w * kSplitInv2), c1 * rne, 0.f, 0.f);
        *reinterpret_cast<float4*>(rs + 4) = make_float4(inv_m1 * (beta * cs.z + cs.y * alpha * xo), per, dwv, dbv);
        if (p.per) p.per[gr] = per;
    }
}

// k_rows for N <= 256: 16 lanes per row, 4 rows per wave (a wave per row is latency-bound: one short dependent chain
// per wave and 655 k waves per launch at cfg4).  Lane l of a row holds the slots 64 c + 4 l .. + 3 of up to four chunks.
__global__ __launch_bounds__(256) void ge2e_tiled_rows16(Problem p, TiledWs L) {
    const int l16 = threadIdx.x & 15;
    const size_t gr_ = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // global row
    const int N = p.N, M = p.M, NM = N * M, npad = L.npad;
    const size_t rows_total = (size_t)p.B * NM;
    const bool live = gr_ < rows_total;
    const size_t gr = live ? gr_ : rows_total - 1;      // dead rows recompute the last row and store nothing
    const int bi = (int)(gr / NM), r = (int)(gr - (size_t)bi * NM), j = r / M;
    const float w = p.w ? *p.w : p.w_imm, bias = p.b ? *p.b : p.b_imm;
    const float eps = p.eps, log_eps = p.log_eps;
    const float inv_m1 = 1.0f / (float)(M - 1);
    const float* X = p.ws + L.x + ((size_t)bi * NM + r) * npad;
    const float4 rst = *reinterpret_cast<const float4*>(p.ws + L.rst + gr * 4);      // rne ke ee
    const float4 cs = *reinterpret_cast<const float4*>(p.ws + L.cst + ((size_t)bi * N + j) * 4);  // rn kap |s| |s|^2
    const float rne = rst.x, ke = rst.y, ee = rst.z;
    const float xo = X[j];
    const float es = xo * cs.z / rne;
    const float eu = (es - ee) * inv_m1;
    const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
    float rnu, ku;
    unit_stats_fast(uu, p.eps_cos, rnu, ku);
    const float cosd = eu * rne * rnu;
    const float sjj = w * (cosd + eps) + bias;
    float c0[4][4], g[4][4];
    float mx = -INFINITY, best = -INFINITY;
    int besti = 0x7fffffff;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kb = 64 * c + 4 * l16;
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb < npad) xv = *reinterpret_cast<const float4*>(X + kb);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kb + e;
            c0[c][e] = (k == j) ? cosd : xe[e];
            if (k < N) {
                const float sv = w * (c0[c][e] + eps) + bias;
                mx = fmaxf(mx, sv);
                if (k != j && sv > best) { best = sv; besti = k; }
            }
        }
    }
    float per;
    if (p.variant == 0) {
        mx = fmaxf(row16_max(mx), log_eps);
        float zoff = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 64 * c + 4 * l16 + e;
                g[c][e] = (k < N) ? __expf(w * (c0[c][e] + eps) + bias - mx) : 0.f;
                if (k != j) zoff += g[c][e];
            }
        zoff = row16_sum(zoff) + __expf(log_eps - mx);
        const float z = zoff + __expf(sjj - mx);
        per = (mx - sjj) + __logf(z);
        const float rz = 1.0f / z;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) g[c][e] = (64 * c + 4 * l16 + e == j) ? -zoff * rz : g[c][e] * rz;
    } else {
        row16_argmax(best, besti);
        const float pos = 1.0f / (1.0f + __expf(-sjj));
        const float neg = (N > 1) ? 1.0f / (1.0f + __expf(-best)) : 0.0f;
        per = 1.0f - pos + neg;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 64 * c + 4 * l16 + e;
                g[c][e] = (k == j) ? -pos * (1.0f - pos) : ((k == besti) ? neg * (1.0f - neg) : 0.f);
            }
    }
    float dwv = 0.f, dbv = 0.f, coef = 0.f, ad = 0.f;
    const float rho_ = rnu * inv_m1, t1_ = ku * cosd * rho_;
    const float own_o = rho_ * (rne + t1_) * cs.z / rne;     // o / (dL/dS on the own column)
    _Float16* GHh = reinterpret_cast<_Float16*>(p.ws + L.gh) + (size_t)bi * 2 * NM * npad + (size_t)r * npad;
    _Float16* GHl = GHh + (size_t)NM * npad;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int kb = 64 * c + 4 * l16;
        if (kb < npad) {
            float gv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                gv[e] = (k < N) ? g[c][e] : 0.f;
                dwv += gv[e] * (c0[c][e] + eps);
                dbv += gv[e];
                coef += gv[e] * c0[c][e];
                // the own-speaker column carries o = c2 |s_j| / (ra w): k_ge's contraction then adds the c2 s_j term of dE
                // by itself, and what o adds to gC_j is, pushed through the centroid norm in k_spk, exactly the
                // leave-one-out speaker row sum_i c3_i e-hat_i minus kap_j (sum_i c3_i xo_i) c-hat_j (ge2e_team.hip, S)
                if (k == j) { ad = gv[e]; gv[e] = own_o * gv[e]; }
            }
            h4 hi, lo;
            split4(make_float4(gv[0] * kSplitScale, gv[1] * kSplitScale, gv[2] * kSplitScale, gv[3] * kSplitScale), hi, lo);
            if (live && p.dE) {      // (forward-only calls: nobody reads the dL/dS planes)
                *reinterpret_cast<h4*>(GHh + kb) = hi;
                *reinterpret_cast<h4*>(GHl + kb) = lo;
            }
        }
    }
    dwv = row16_sum(dwv); dbv = row16_sum(dbv);
    coef = w * row16_sum(coef); ad = w * row16_sum(ad);
    if (l16 == 0 && live) {
        const float rho = rnu * inv_m1;
        const float c2 = rho * (ad * rne + ad * ku * cosd * rnu * inv_m1);
        const float c1 = (-ke * coef * rne - ad * rnu * inv_m1) - c2 / rne;
        const float alpha = ad * rnu * (1.0f + ku * cosd * rho / rne);
        const float beta = -ad * rnu * ku * cosd * rho;
        float* rs = p.ws + L.rs + gr * 8;
        // ra, c1e; then the row's share of the c-hat_j coefficient of KJ_j: c4' = (beta |s_j| + kap_j alpha xo) / (M - 1)
        *reinterpret_cast<float4*>(rs) = make_float4(rne * (w * kSplitInv2), c1 * rne, 0.f, 0.f);
        *reinterpret_cast<float4*>(rs + 4) = make_float4(inv_m1 * (beta * cs.z + cs.y * alpha * xo), per, dwv, dbv);
        if (p.per) p.per[gr] = per;
    }
}

// ---------------------------------------------------------------------------------------------
// k_gc: gC[k][d] = sum_r GH[r][k] EH[r][d].  One workgroup per (TM centroids x TN d) tile over all rows of the batch
// (both operands with K along their rows -> transposing reads).
template <class C>
__global__ __launch_bounds__(C::NT, 2) void ge2e_tiled_gc(Problem p, TiledWs L) {
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = p.N, D = p.D, NM = p.N * p.M, npad = L.npad;
    const int dtiles = (D + C::TN - 1) / C::TN, ct = (N + C::TM - 1) / C::TM;
    const int S = L.gc_split, steps = NM / 32;
    struct Ix { int dt, kt, sp, bi; };
    auto decode = [&](int t, Opnd& A, Opnd& Bo, Ix& ix, int& ktotal) {
        ix.dt = t % dtiles; t /= dtiles;
        ix.kt = t % ct; t /= ct;
        ix.sp = t % S;
        ix.bi = t / S;
        // rows [r0, r1) of the batch: piece sp of S (whole 32-row K-steps when S > 1 -- the launcher splits only then)
        const int r0 = S > 1 ? (int)((long long)steps * ix.sp / S) * 32 : 0;
        const int r1 = S > 1 ? (int)((long long)steps * (ix.sp + 1) / S) * 32 : NM;
        A.hi = reinterpret_cast<const _Float16*>(p.ws + L.gh) + (size_t)ix.bi * 2 * NM * npad + (size_t)r0 * npad + ix.kt * C::TM;
        A.lo = A.hi + (size_t)NM * npad;
        A.ld = npad; A.valid = min(C::TM, npad - ix.kt * C::TM);
        Bo.hi = reinterpret_cast<const _Float16*>(p.ws + L.eh) + (size_t)ix.bi * 2 * NM * D + (size_t)r0 * D + ix.dt * C::TN;
        Bo.lo = Bo.hi + (size_t)NM * D;
        Bo.ld = D; Bo.valid = min(C::TN, D - ix.dt * C::TN);
        ktotal = r1 - r0;
    };
    GE2E_PROF_DECL(8)
    unsigned long long t_first[5] = {0, 0, 0, 0, 0};
    auto epilogue = [&](const Ix& ix, f32x16 (&acc)[C::A2][C::B2]) {
        GE2E_PROF_AT(0, t_first[0]);
        GE2E_PROF(1);
        float* GC = S > 1 ? p.ws + L.x + ((size_t)ix.bi * S + ix.sp) * N * D : p.ws + L.gc + (size_t)ix.bi * N * D;
        const int l31 = lane & 31, h = lane >> 5, wa = wid / C::WN, wb = wid % C::WN, pq = lane & 3, cq = l31 >> 2;
#pragma unroll
        for (int a2 = 0; a2 < C::A2; ++a2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int k = ix.kt * C::TM + wa * (32 * C::A2) + 32 * a2 + 8 * g + 4 * h + pq;
#pragma unroll
                for (int b2 = 0; b2 < C::B2; ++b2) {
                    float x[4] = {acc[a2][b2][4 * g], acc[a2][b2][4 * g + 1], acc[a2][b2][4 * g + 2], acc[a2][b2][4 * g + 3]};
                    quad_transpose4(x, lane);
                    const int d = ix.dt * C::TN + wb * (32 * C::B2) + 32 * b2 + 4 * cq;
                    if (k < N && d < D)
                        *reinterpret_cast<float4*>(GC + (size_t)k * D + d) =
                            make_float4(x[0] * kSplitInv2, x[1] * kSplitInv2, x[2] * kSplitInv2, x[3] * kSplitInv2);
                }
            }
        GE2E_PROF(2);
    };
    walk_tiles<C, false, false, Ix>(p.B * S * ct * dtiles, gsm, tid, decode, epilogue, t_first);
#ifdef GE2E_PROFILE
    for (int i = 0; i < 4; ++i) prof_acc[4 + i] = t_first[1 + i];
#endif
    GE2E_PROF_FLUSH_AT(8, 8)
}

// ---------------------------------------------------------------------------------------------
// k_spk: one wave per speaker: dc_j / M (gC through the centroid norm) + the leave-one-out sums.
__global__ __launch_bounds__(256) void ge2e_tiled_spk(Problem p, TiledWs L) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = p.N, M = p.M, D = p.D, NM = N * M;
    if (gw >= p.B * N) return;
    const int bi = gw / N, j = gw - bi * N;
    const float w = p.w ? *p.w : p.w_imm;
    const float fM = (float)M;
    const int S = L.gc_split;
    const float* GC = S > 1 ? p.ws + L.x + ((size_t)bi * S * N + j) * D : p.ws + L.gc + ((size_t)bi * N + j) * D;
    const float* CHf = p.ws + L.chf + ((size_t)bi * N + j) * D;
    const float4 cs = *reinterpret_cast<const float4*>(p.ws + L.cst + ((size_t)bi * N + j) * 4);
    const float* RS = p.ws + L.rs + ((size_t)bi * NM + (size_t)j * M) * 8;
    float* KJ = p.ws + L.kj + ((size_t)bi * N + j) * D;
    const int npass = (D + 255) >> 8;
    float4 g[4], c[4];
    float coef = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = 256 * q + 4 * lane;
        g[q] = c[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < npass && d < D) {
            float4 gs = *reinterpret_cast<const float4*>(GC + d);
            for (int sp = 1; sp < S; ++sp) {       // k_gc's partial sums, in a fixed order
                const float4 gp = *reinterpret_cast<const float4*>(GC + (size_t)sp * N * D + d);
                gs.x += gp.x; gs.y += gp.y; gs.z += gp.z; gs.w += gp.w;
            }
            g[q] = scale4(gs, w);
            c[q] = *reinterpret_cast<const float4*>(CHf + d);
            coef += dot4(g[q], c[q]);
        }
    }
    coef = wave_sum(coef);
    const float f = cs.y * coef, sc = cs.x / fM;
    // sum_i c4'_i: the whole speaker row KJP_j is a multiple of c-hat_j (the e-hat part rides in gC).  Lane i reads row i's
    // coefficient and the wave adds them (as a loop over i this was M dependent round trips per wave: a run-time M is not unrolled)
    float bsum = 0.f;
    for (int i = lane; i < M; i += kWave) bsum += RS[i * 8 + 4];
    const float bs = wave_sum(bsum);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = 256 * q + 4 * lane;
        if (q < npass && d < D) {
            *reinterpret_cast<float4*>(KJ + d) =
                make_float4((g[q].x - f * c[q].x) * sc + bs * c[q].x, (g[q].y - f * c[q].y) * sc + bs * c[q].y,
                            (g[q].z - f * c[q].z) * sc + bs * c[q].z, (g[q].w - f * c[q].w) * sc + bs * c[q].w);
        }
    }
    // the batch's loss / dw / db: fixed-order sums of the per-row values (k_rows wrote them two launches ago), taken by
    // the wave of speaker 0 -- one launch fewer than a reduction kernel of its own (forward-only calls keep k_reduce)
    if (j == 0) {
        const float* RSb = p.ws + L.rs + (size_t)bi * NM * 8;
        float red[3] = {0.f, 0.f, 0.f};
#pragma unroll 8     // (eight loads in flight: this wave has NM / 64 of them in a row while its 4 N - 1 siblings have one round trip)
        for (int r = lane; r < NM; r += kWave) {
            const float4 v = *reinterpret_cast<const float4*>(RSb + (size_t)r * 8 + 4);   // . loss dw db
            red[0] += v.y; red[1] += v.z; red[2] += v.w;
        }
        wave_sum_n<3>(red);
        if (lane == 0) {
            if (p.loss) p.loss[bi] = red[0];
            if (p.dw) p.dw[bi] = red[1];
            if (p.db) p.db[bi] = red[2];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_ge: gE[r][d] = sum_k GH[r][k] CH[k][d] per (TM rows x TN d) tile (A K-contiguous, B K-rows), then the epilogue
// dE = ra gE + c1e e + KJ_j (the c2 s_j term rides in GH's own-speaker column).
template <class C>
__global__ __launch_bounds__(C::NT, 2) void ge2e_tiled_ge(Problem p, TiledWs L) {
    extern __shared__ __attribute__((aligned(16))) _Float16 gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int N = p.N, M = p.M, D = p.D, NM = N * M, npad = L.npad;
    const int dtiles = (D + C::TN - 1) / C::TN, rt = (NM + C::TM - 1) / C::TM;
    struct Ix { int dt, rtile, bi; };
    auto decode = [&](int t, Opnd& A, Opnd& Bo, Ix& ix, int& ktotal) {
        ix.dt = t % dtiles; t /= dtiles;
        ix.rtile = t % rt;
        ix.bi = t / rt;
        A.hi = reinterpret_cast<const _Float16*>(p.ws + L.gh) + (size_t)ix.bi * 2 * NM * npad + (size_t)ix.rtile * C::TM * npad;
        A.lo = A.hi + (size_t)NM * npad;
        A.ld = npad; A.valid = min(C::TM, NM - ix.rtile * C::TM);
        Bo.hi = reinterpret_cast<const _Float16*>(p.ws + L.ch) + (size_t)ix.bi * 2 * N * D + ix.dt * C::TN;
        Bo.lo = Bo.hi + (size_t)N * D;
        Bo.ld = D; Bo.valid = min(C::TN, D - ix.dt * C::TN);
        ktotal = N;      // K = the N real centroid slots (pad columns of GH are zero)
    };
    GE2E_PROF_DECL(8)
    unsigned long long t_first[5] = {0, 0, 0, 0, 0};
    auto epilogue = [&](const Ix& ix, f32x16 (&acc)[C::A2][C::B2]) {
        GE2E_PROF_AT(0, t_first[0]);
        GE2E_PROF(1);
        const int rtile = ix.rtile, dt = ix.dt;
        const float* E = p.E + (size_t)ix.bi * NM * D;
        float* dE = p.dE + (size_t)ix.bi * NM * D;
        const float* KJ = p.ws + L.kj + (size_t)ix.bi * N * D;
        const float* RS = p.ws + L.rs + (size_t)ix.bi * NM * 8;
        // in-quad transposes turn four accumulator registers (4 rows x this lane's column) into one row x 4 consecutive
        // columns: every global access of the epilogue is 16 bytes wide (8 rows x 128 B per wave-instruction)
        const int l31 = lane & 31, h = lane >> 5, wa = wid / C::WN, wb = wid % C::WN, pq = lane & 3, cq = l31 >> 2;
        // The loads of a 32-row block (4 row scalars, 8 x 16 B of e, 8 x 16 B of KJ per lane: 72 VGPRs, free now that the
        // fragments are dead) are all requested before the first is used, on clamped addresses so that no branch separates
        // them: one memory round trip per block instead of one per 16 bytes (stamped build, config 5, round 4: the epilogue
        // was 35 % of the workgroup's time at 12 GB/s per CU -- latency, not bandwidth).
#pragma unroll
        for (int a2 = 0; a2 < C::A2; ++a2) {
            float2 rs[4];
            float4 e[4][C::B2], kj[4][C::B2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = rtile * C::TM + wa * (32 * C::A2) + 32 * a2 + 8 * g + 4 * h + pq;
                const int rc = r < NM ? r : NM - 1;
                rs[g] = *reinterpret_cast<const float2*>(RS + (size_t)rc * 8);  // ra c1e
                const int j = rc / M;
#pragma unroll
                for (int b2 = 0; b2 < C::B2; ++b2) {
                    const int d = dt * C::TN + wb * (32 * C::B2) + 32 * b2 + 4 * cq;
                    const int dc = d < D ? d : 0;
                    e[g][b2] = *reinterpret_cast<const float4*>(E + (size_t)rc * D + dc);
                    kj[g][b2] = *reinterpret_cast<const float4*>(KJ + (size_t)j * D + dc);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = rtile * C::TM + wa * (32 * C::A2) + 32 * a2 + 8 * g + 4 * h + pq;
#pragma unroll
                for (int b2 = 0; b2 < C::B2; ++b2) {
                    float x[4] = {acc[a2][b2][4 * g], acc[a2][b2][4 * g + 1], acc[a2][b2][4 * g + 2], acc[a2][b2][4 * g + 3]};
                    quad_transpose4(x, lane);
                    const int d = dt * C::TN + wb * (32 * C::B2) + 32 * b2 + 4 * cq;
                    if (r < NM && d < D)
                        *reinterpret_cast<float4*>(dE + (size_t)r * D + d) =
                            make_float4(x[0] * rs[g].x + e[g][b2].x * rs[g].y + kj[g][b2].x, x[1] * rs[g].x + e[g][b2].y * rs[g].y + kj[g][b2].y,
                                        x[2] * rs[g].x + e[g][b2].z * rs[g].y + kj[g][b2].z, x[3] * rs[g].x + e[g][b2].w * rs[g].y + kj[g][b2].w);
                }
            }
        }
        GE2E_PROF(2);
    };
    walk_tiles<C, true, false, Ix>(p.B * rt * dtiles, gsm, tid, decode, epilogue, t_first);
#ifdef GE2E_PROFILE
    for (int i = 0; i < 4; ++i) prof_acc[4 + i] = t_first[1 + i];
#endif
    GE2E_PROF_FLUSH_AT(16, 8)
}

// ---------------------------------------------------------------------------------------------
// k_reduce: one workgroup per batch, fixed-order sums of the per-row loss / dw / db.
// get_cos_sim's output from the similarity contraction (s3:42-80): cos[r][k] = X[r][k] + eps, the own column replaced by the
// cosine with the leave-one-out centroid, which follows from X's own column and the row / speaker scalars of the
// preparation pass exactly as in the row kernels above.  One wave per row, lanes along the slots: coalesced both ways.
__global__ __launch_bounds__(256) void ge2e_tiled_cos(Problem p, TiledWs L) {
    const int lane = threadIdx.x & 63;
    const size_t gr = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // global row
    const int N = p.N, M = p.M, NM = N * M, npad = L.npad;
    if (gr >= (size_t)p.B * NM) return;
    const int bi = (int)(gr / NM), r = (int)(gr - (size_t)bi * NM), j = r / M;
    const float* X = p.ws + L.x + ((size_t)bi * NM + r) * npad;
    const float4 rst = *reinterpret_cast<const float4*>(p.ws + L.rst + gr * 4);                   // rne ke ee
    const float4 cs = *reinterpret_cast<const float4*>(p.ws + L.cst + ((size_t)bi * N + j) * 4);  // rn kap |s| |s|^2
    const float inv_m1 = 1.0f / (float)(M - 1);
    const float rne = rst.x, ee = rst.z;
    const float es = X[j] * cs.z / rne;
    const float eu = (es - ee) * inv_m1;
    const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
    float rnu, ku;
    unit_stats_fast(uu, p.eps_cos, rnu, ku);
    const float cosd = eu * rne * rnu;
    float* out = p.cos_out + gr * N;
    for (int k = lane; k < N; k += 64) out[k] = (k == j ? cosd : X[k]) + p.eps;
}

__global__ __launch_bounds__(256) void ge2e_tiled_reduce(Problem p, TiledWs L) {
    __shared__ float red[3][256];
    const int bi = blockIdx.x, tid = threadIdx.x, NM = p.N * p.M;
    const float* RS = p.ws + L.rs + (size_t)bi * NM * 8;
    float l = 0.f, a = 0.f, c = 0.f;
    for (int r = tid; r < NM; r += 256) { l += RS[(size_t)r * 8 + 5]; a += RS[(size_t)r * 8 + 6]; c += RS[(size_t)r * 8 + 7]; }
    red[0][tid] = l; red[1][tid] = a; red[2][tid] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { red[0][tid] += red[0][tid + s]; red[1][tid] += red[1][tid + s]; red[2][tid] += red[2][tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        if (p.loss) p.loss[bi] = red[0][0];
        if (p.dw) p.dw[bi] = red[1][0];
        if (p.db) p.db[bi] = red[2][0];
    }
}

// ---------------------------------------------------------------------------------------------
bool tiled_supports(int N, int M, int D) {
    return N >= 1 && N <= 1024 && M >= 2 && D >= 8 && D <= 1024 && (D % 8) == 0;     // rows of the fp16 planes 16-byte aligned
}

TiledWs tiled_layout(int B, int N, int M, int D) {
    TiledWs L;
    const size_t NM = (size_t)N * M;
    L.npad = (N + 63) / 64 * 64;
    L.row_tiles = (int)((NM + 63) / 64);
    L.cen_tiles = L.npad / 64;
    size_t off = 0;  // offsets in floats; fp16 planes take half a float per element
    auto take = [&](size_t floats) { const size_t o = off; off = align_up(off + floats, 64); return o; };
    L.ch = take((size_t)B * N * D);          // 2 planes x [N][D] halfs
    L.eh = take((size_t)B * NM * D);         // 2 planes x [NM][D] halfs
    L.gh = take((size_t)B * NM * L.npad);    // 2 planes x [NM][npad] halfs
    L.chf = take((size_t)B * N * D);
    L.x = take((size_t)B * NM * L.npad);
    L.gc = take((size_t)B * N * D);
    L.kj = take((size_t)B * N * D);
    L.cst = take((size_t)B * N * 4);
    L.rst = take((size_t)B * NM * 4);
    L.rs = take((size_t)B * NM * 8);
    L.total = off;
    return L;
}

size_t tiled_workspace_bytes(int B, int N, int M, int D) { return tiled_layout(B, N, M, D).total * sizeof(float); }

hipError_t launch_tiled(const Problem& p, hipStream_t stream) {
    TiledWs L = tiled_layout(p.B, p.N, p.M, p.D);
    const int NM = p.N * p.M;
    const unsigned spk_blocks = (unsigned)((p.B * p.N + 3) / 4);
    const unsigned row_blocks = (unsigned)(((size_t)p.B * NM + 3) / 4);
    typedef GemmCfg<128, 128> C1;
    typedef GemmCfg<256, 256> C2;
    typedef GemmCfgDma C3;   // the same tile, operands by LDS-DMA: the contraction's K must be a multiple of 32
    {   // every launch, like the fused kernels: the attribute is per device and a process may drive several
        const void* small[] = {reinterpret_cast<const void*>(ge2e_tiled_sim<C1>), reinterpret_cast<const void*>(ge2e_tiled_gc<C1>),
                               reinterpret_cast<const void*>(ge2e_tiled_ge<C1>)};
        const void* big[] = {reinterpret_cast<const void*>(ge2e_tiled_sim<C2>), reinterpret_cast<const void*>(ge2e_tiled_gc<C2>),
                             reinterpret_cast<const void*>(ge2e_tiled_ge<C2>),
                             reinterpret_cast<const void*>(ge2e_tiled_sim<C3>), reinterpret_cast<const void*>(ge2e_tiled_gc<C3>),
                             reinterpret_cast<const void*>(ge2e_tiled_ge<C3>),
                             reinterpret_cast<const void*>(ge2e_tiled_simrows<C3, false, false>), reinterpret_cast<const void*>(ge2e_tiled_simrows<C3, false, true>),
                             reinterpret_cast<const void*>(ge2e_tiled_simrows<C3, true, false>), reinterpret_cast<const void*>(ge2e_tiled_simrows<C3, true, true>)};
        for (const void* fn : small) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        for (const void* fn : big) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
    }
    auto tiles = [](int n, int t) { return (unsigned)((n + t - 1) / t); };
    // the DMA-fed contractions walk their tiles with one workgroup per CU (160 KB of LDS: one fits)
    int ncu = device_cu_count();
    if (ncu < 8) ncu = 8;
    ncu = ncu / 8 * 8;    // a workgroup's tiles b, b + grid, ... stay on one XCD when the grid is a multiple of 8
    auto walk_grid = [&](unsigned ntiles) { return dim3(ntiles < (unsigned)ncu ? ntiles : (unsigned)ncu); };
    // the 256 x 256 tile where both extents of the contraction's output reach it AND its grid still gives every CU
    // most of the CUs a workgroup (a single batch of cfg5 has 160 big tiles), else 128 x 128
    const unsigned fill = 192;   // (measured: the long-K gC contraction gains from the big tile even at one workgroup per CU)
    const bool big_sim = NM >= 256 && p.N >= 256 && (unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256) >= fill;
    const bool big_gc = p.N >= 256 && p.D >= 256 && (unsigned)p.B * tiles(p.N, 256) * tiles(p.D, 256) >= fill;
    const bool big_ge = NM >= 256 && p.D >= 256 && (unsigned)p.B * tiles(NM, 256) * tiles(p.D, 256) >= fill;
    // one 256-slot tile holds a whole similarity row: similarity contraction + row pass in one kernel (config 4)
    const bool fused_rows = L.npad <= 256 && p.N > 128 && NM >= 256 && p.D % 32 == 0 && (unsigned)p.B * tiles(NM, 256) >= fill;
    // k_gc has B ct dtiles tiles, each over ALL rows of its batch: at config 5 that is 192 workgroups for 256 CUs, one
    // round.  Cut the rows into S pieces when that fills the last round better (S = 4 there: three full rounds); the
    // partial sums go to the similarity block, dead once the row pass has run, and k_spk adds them in a fixed order.
    if (big_gc && NM % 32 == 0) {
        const unsigned tiles_gc = (unsigned)p.B * tiles(p.N, 256) * tiles(p.D, 256);
        auto fillq = [](unsigned wg) { return (double)wg / (double)((wg + 255) / 256 * 256); };
        int best = 1;
        for (int sp : {2, 4, 8})
            if (NM / 32 >= 8 * sp && (size_t)sp * p.N * p.D <= (size_t)NM * L.npad && fillq(tiles_gc * sp) > fillq(tiles_gc * best) + 0.1) best = sp;
        L.gc_split = best;
    }
    launch_prep(p, L, stream);
    if (fused_rows) {
        const dim3 g((unsigned)p.B * tiles(NM, 256));
        if (p.variant == 1) {
            if (p.N == 256) hipLaunchKernelGGL((ge2e_tiled_simrows<C3, true, true>), g, dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
            else hipLaunchKernelGGL((ge2e_tiled_simrows<C3, true, false>), g, dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        } else {
            if (p.N == 256) hipLaunchKernelGGL((ge2e_tiled_simrows<C3, false, true>), g, dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
            else hipLaunchKernelGGL((ge2e_tiled_simrows<C3, false, false>), g, dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        }
    } else if (big_sim && p.D % 32 == 0)
        hipLaunchKernelGGL(ge2e_tiled_sim<C3>, walk_grid((unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
    else if (big_sim)
        hipLaunchKernelGGL(ge2e_tiled_sim<C2>, dim3((unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
    else
        hipLaunchKernelGGL(ge2e_tiled_sim<C1>, dim3((unsigned)p.B * tiles(NM, 128) * tiles(p.N, 128)), dim3(C1::NT), C1::LDS_BYTES, stream, p, L);
    if (fused_rows) {
    } else if (L.npad <= 256)
        hipLaunchKernelGGL(ge2e_tiled_rows16, dim3((unsigned)(((size_t)p.B * NM + 15) / 16)), dim3(256), 0, stream, p, L);
    else
        hipLaunchKernelGGL(ge2e_tiled_rows, dim3(row_blocks), dim3(256), 0, stream, p, L);
    if (p.dE) {
        if (big_gc && NM % 32 == 0)
            hipLaunchKernelGGL(ge2e_tiled_gc<C3>, walk_grid((unsigned)p.B * tiles(p.N, 256) * tiles(p.D, 256) * L.gc_split), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        else if (big_gc)
            hipLaunchKernelGGL(ge2e_tiled_gc<C2>, dim3((unsigned)p.B * tiles(p.N, 256) * tiles(p.D, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        else
            hipLaunchKernelGGL(ge2e_tiled_gc<C1>, dim3((unsigned)p.B * tiles(p.N, 128) * tiles(p.D, 128)), dim3(C1::NT), C1::LDS_BYTES, stream, p, L);
        hipLaunchKernelGGL(ge2e_tiled_spk, dim3(spk_blocks), dim3(256), 0, stream, p, L);
        if (big_ge && p.N % 32 == 0)
            hipLaunchKernelGGL(ge2e_tiled_ge<C3>, walk_grid((unsigned)p.B * tiles(NM, 256) * tiles(p.D, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        else if (big_ge)
            hipLaunchKernelGGL(ge2e_tiled_ge<C2>, dim3((unsigned)p.B * tiles(NM, 256) * tiles(p.D, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
        else
            hipLaunchKernelGGL(ge2e_tiled_ge<C1>, dim3((unsigned)p.B * tiles(NM, 128) * tiles(p.D, 128)), dim3(C1::NT), C1::LDS_BYTES, stream, p, L);
    }
    else   // forward only: nothing ran after k_rows that could carry the sums
        hipLaunchKernelGGL(ge2e_tiled_reduce, dim3((unsigned)p.B), dim3(256), 0, stream, p, L);
    return hipGetLastError();
}

// Forward half only, for ge2e_cos_sim: preparation, similarity contraction on the matrix cores, cos output.
hipError_t launch_tiled_cos(const Problem& p, hipStream_t stream) {
    const TiledWs L = tiled_layout(p.B, p.N, p.M, p.D);
    const int NM = p.N * p.M;
    typedef GemmCfg<128, 128> C1;
    typedef GemmCfg<256, 256> C2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_tiled_sim<C1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C1::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_tiled_sim<C2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_tiled_sim<GemmCfgDma>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C2::LDS_BYTES);
    if (e != hipSuccess) return e;
    auto tiles = [](int n, int t) { return (unsigned)((n + t - 1) / t); };
    const bool big_sim = NM >= 256 && p.N >= 256 && (unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256) >= 192;
    launch_prep(p, L, stream);
    if (big_sim && p.D % 32 == 0)
    {   // (one workgroup per CU walks the tiles: see launch_tiled)
        const int ncu = device_cu_count() < 8 ? 8 : device_cu_count();
        const unsigned nt = (unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256), cap = (unsigned)(ncu / 8 * 8);
        hipLaunchKernelGGL(ge2e_tiled_sim<GemmCfgDma>, dim3(nt < cap ? nt : cap), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
    }
    else if (big_sim)
        hipLaunchKernelGGL(ge2e_tiled_sim<C2>, dim3((unsigned)p.B * tiles(NM, 256) * tiles(p.N, 256)), dim3(C2::NT), C2::LDS_BYTES, stream, p, L);
    else
        hipLaunchKernelGGL(ge2e_tiled_sim<C1>, dim3((unsigned)p.B * tiles(NM, 128) * tiles(p.N, 128)), dim3(C1::NT), C1::LDS_BYTES, stream, p, L);
    hipLaunchKernelGGL(ge2e_tiled_cos, dim3((unsigned)(((size_t)p.B * NM + 3) / 4)), dim3(256), 0, stream, p, L);
    return hipGetLastError();
}

}  // namespace ge2e
