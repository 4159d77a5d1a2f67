// Launchers of the static helpers' backward passes (ge2e_helpers.hip).
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

hipError_t launch_utt_centroids(const float* E, int B, int N, int M, int D, float* U, hipStream_t stream);
hipError_t launch_centroids_bwd(const float* g, int B, int N, int M, int D, float* dE, hipStream_t stream);
// n local speakers whose own columns are j0 .. j0 + n - 1 of the N centroids (n = N, j0 = 0: the whole batch)
size_t cos_bwd_workspace_bytes(int B, int n, int N, int M, int D);
hipError_t launch_cos_bwd(const float* E, const float* C, const float* cosv, const float* gcos, int B, int n, int N, int j0,
                          int M, int D, float eps_cos, float eps, float* dE, float* dC, float* ws, hipStream_t stream);
hipError_t launch_calc_loss_bwd(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant,
                                const float* gloss, const float* gper, float* dS, hipStream_t stream);

hipError_t launch_scale_grads(const float* dE, const float* dw, const float* db, const float* g, int gB, int B,
                              size_t per_batch, float* gE, float* gw, float* gb, hipStream_t stream);

}  // namespace ge2e
