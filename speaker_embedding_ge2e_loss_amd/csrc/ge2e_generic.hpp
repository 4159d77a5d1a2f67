// Workspace layout + launchers of GE2E_IMPL_GENERIC (see ge2e_generic.hip).
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

// Per-workgroup workspace slice, offsets in floats.
struct GenericLayout {
    size_t ch, cht, ss, gc, a, rowstat, cstat, total;
    int npad;
};

__host__ __device__ inline GenericLayout generic_layout(int N, int M, int D) {
    GenericLayout L;
    L.npad = (N + 63) / 64 * 64;
    const size_t nd = (size_t)N * D;
    L.ch = 0;
    L.cht = L.ch + nd;
    L.ss = L.cht + (size_t)D * L.npad;
    L.gc = L.ss + nd;
    L.a = L.gc + nd;
    L.rowstat = L.a + (size_t)N * M * N;
    L.cstat = L.rowstat + (size_t)N * M * 8;
    L.total = align_up(L.cstat + (size_t)N * 4, 64);
    return L;
}

int generic_grid(int B);
size_t generic_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_generic(const Problem& p, hipStream_t stream);
// n local speakers whose columns are j0 .. j0 + n - 1 of N (n = N, j0 = 0: the whole batch)
hipError_t launch_calc_loss(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, float* loss,
                            float* per, hipStream_t stream);
hipError_t launch_cos_centroids(const float* E, const float* C, int B, int n, int N, int j0, int M, int D, float eps_cos,
                                float eps, float* cos, hipStream_t stream);
hipError_t launch_centroids(const float* E, int B, int N, int M, int D, float* cent, hipStream_t stream);

}  // namespace ge2e
