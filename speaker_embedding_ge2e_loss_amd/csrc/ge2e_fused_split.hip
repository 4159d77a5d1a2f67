// GE2E_IMPL_FUSED_SPLIT: one workgroup per batch, three sweeps over E like ge2e_fused_f32.hip
// (read that file's header first), with two structural differences:
//
//  1. The three contractions run on the fp16 matrix cores at fp32-grade accuracy: operands live
//     in LDS as fp16 hi + lo images and every product is hi.hi + hi.lo + lo.hi in an fp32
//     accumulator (ge2e_split_gemm.hpp) -- 3/16 of the cost of v_mfma_f32_32x32x2_f32.  The same
//     image serves a contraction along its rows (ds_read_b128 fragments: X = CH . ET^T) and along
//     its columns (ds_read_b64_tr_b16 fragments: gE = G . CH, gC = G^T . ET).  Everything outside
//     the contractions (norms, softmax, gradient algebra, epilogue) is fp32 on the VALU.
//  2. 512 threads = 8 waves = two per SIMD.  Once the matrix work shrank 5x the kernel was bound by
//     VALU phases issuing from a single wave per SIMD (one instruction per 4 cycles); two waves per
//     SIMD double the vector issue rate and hide each other's LDS / memory latency.  Every VALU
//     phase splits its rows over 8 waves; the gradient contractions give each wave a 32 x 64
//     block (two waves share a SIMD's matrix pipe); X (four 32 x 32 tiles) stays on waves 0-3.
//
//   LDS (D=256: 157 KB)            fp16 images, pitch D+16 halfs (16-B aligned rows, conflict-free for b128 and tr reads)
//     CHh, CHl [64][D+16]   unit centroids * 2^8, hi / lo
//     ETh, ETl [64][D+16]   the tile's unit embeddings * 2^8, hi / lo  (fp32 gC staging in finalize)
//     U        18 KB       union: S [64][68] fp32 (similarities)  |  Gh, Gl [64][72] fp16 (G_off * 2^8)
//                                 |  the epilogue's 8 x [8][68] fp32 staging blocks
//     RS [64][8], CST [64][4] fp32
//
// Leave-one-out statistics come out of the similarity tile instead of extra dot products: the
// own-speaker column of X is c-hat_j . e-hat_r, so  e.s_j = X |s_j| |e|,
//   e.u = (e.s_j - |e|^2)/(M-1),  |u|^2 = (|s_j|^2 - 2 e.s_j + |e|^2)/(M-1)^2.
// (u = 0 exactly, i.e. the other M-1 utterances summing to zero, is where this form loses digits
// against the reference's explicit u; GE2E_IMPL_GENERIC keeps the explicit form.)
// The operand of the gradient contractions is G_off = dL/dS (in [0,1]); w is applied to the
// accumulators.
#include "ge2e_common.hpp"
#include "ge2e_fused.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_fused_split_body.hpp"

namespace ge2e {
using namespace fsplit;

size_t fused_split_lds_bytes(int Dc) {
    const int D = (Dc + 63) / 64 * 64;   // the kernel's column count: the caller's D padded to a multiple of 64
    const int PH = D + 16;
    return (size_t)4 * 64 * PH * 2 + (size_t)2 * 64 * GP * 2 + (size_t)(TR * 8 + NC * 4 + 32) * sizeof(float);
}
template <int NCH>  // D = 64 * NCH
__global__ __launch_bounds__(512, 2) void ge2e_fused_split_kernel(Problem p, FusedWs wsl) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    fsplit::body<NCH>(p, wsl, smem_f, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// D: any multiple of 4 up to 256 (rows 16-byte aligned); the kernel works on ceil(D / 64) * 64 columns, the rest zeros
constexpr int pad64(int D) { return (D + 63) / 64 * 64; }
bool fused_split_supports(int N, int M, int D) {
    return N >= 1 && N <= 64 && M >= 2 && M <= 64 && D >= 4 && D <= 256 && (D % 4) == 0;
}

FusedWs fused_split_layout(int N, int M, int Dc) {
    const int D = pad64(Dc);
    FusedWs L;
    int spt = TR / M;
    if (spt > MAX_SPT) spt = MAX_SPT;
    if (spt > N) spt = N;
    L.spt = spt;
    L.ntiles = (N + spt - 1) / spt;
    L.stash_a = 0;                                             // [ntiles][2][64][64] halfs
    L.stash_rs = L.stash_a + (size_t)L.ntiles * TR * NC;       // [ntiles][64][4] floats
    L.dcm = L.stash_rs + (size_t)L.ntiles * TR * 4;            // [64][D] complete speaker rows KJ
    L.dump = L.dcm + (size_t)NC * D;                           // [64][D] partial speaker rows KJP
    L.sums = L.dump + (size_t)NC * D;                          // [64][D] speaker sums of the NEXT batch
    L.stride = align_up(L.sums + (size_t)NC * D, 64);
    return L;
}

// one persistent workgroup per CU of the current device (fewer did not pay: profiles/r01, grid experiment)
int fused_split_grid(int B) {
    const int cus = device_cu_count();
    return B < cus ? B : cus;
}

size_t fused_split_workspace_bytes(int B, int N, int M, int D) {
    return (size_t)fused_split_grid(B) * fused_split_layout(N, M, D).stride * sizeof(float);
}

template <int NCH>
static hipError_t launch_nch(const Problem& p, const FusedWs& L, size_t lds, hipStream_t stream) {
    static KernelLaunchState state;     // one per instantiation; per-device entries inside
    hipError_t err = prepare_kernel(state, reinterpret_cast<const void*>(ge2e_fused_split_kernel<NCH>), 512, (unsigned)lds, nullptr);
    if (err != hipSuccess) return err;
    int grid = fused_split_grid(p.B);
    if (p.grid_cap > 0 && grid > p.grid_cap) grid = p.grid_cap;
    hipLaunchKernelGGL(ge2e_fused_split_kernel<NCH>, dim3(grid), dim3(512), lds, stream, p, L);
    return hipGetLastError();
}

hipError_t launch_fused_split(const Problem& p, hipStream_t stream) {
    const size_t lds = fused_split_lds_bytes(p.D);
    const FusedWs L = fused_split_layout(p.N, p.M, p.D);
    switch (pad64(p.D) / 64) {
        case 1: return launch_nch<1>(p, L, lds, stream);
        case 2: return launch_nch<2>(p, L, lds, stream);
        case 3: return launch_nch<3>(p, L, lds, stream);
        default: return launch_nch<4>(p, L, lds, stream);
    }
}

}  // namespace ge2e
