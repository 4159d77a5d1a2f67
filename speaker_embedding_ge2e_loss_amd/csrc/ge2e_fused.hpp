// Workspace layout + launchers of the fused one-workgroup-per-batch kernels.
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

// Per-workgroup workspace slice (offsets in floats) and the tiling of a batch.
struct FusedWs {
    int spt;       // whole speakers per 64-row tile
    int ntiles;    // tiles per batch
    size_t stash_a, stash_rs, dcm, dump, sums, stride;
};

bool fused_f32_supports(int N, int M, int D);
FusedWs fused_f32_layout(int N, int M, int D);
int fused_f32_grid(int B);
size_t fused_f32_lds_bytes(int D);
size_t fused_f32_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_fused_f32(const Problem& p, hipStream_t stream);


bool fused_split_supports(int N, int M, int D);
FusedWs fused_split_layout(int N, int M, int D);
int fused_split_grid(int B);
size_t fused_split_lds_bytes(int D);
size_t fused_split_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_fused_split(const Problem& p, hipStream_t stream);

}  // namespace ge2e
