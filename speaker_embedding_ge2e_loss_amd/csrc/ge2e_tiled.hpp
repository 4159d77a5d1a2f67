// Workspace layout + launcher of GE2E_IMPL_TILED (see ge2e_tiled.hip).
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

struct TiledWs {  // offsets in floats into the workspace
    size_t ch, eh, gh, chf, x, gc, kj, cst, rst, rs, total;
    int npad, row_tiles, cen_tiles;
    int gc_split = 1;   // k_gc's contraction over the rows cut into this many pieces (partial sums in the dead X block; k_spk adds them)
};

bool tiled_supports(int N, int M, int D);
TiledWs tiled_layout(int B, int N, int M, int D);
size_t tiled_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_tiled(const Problem& p, hipStream_t stream);
hipError_t launch_tiled_cos(const Problem& p, hipStream_t stream);   // p.cos_out = get_cos_sim(E): forward half only

}  // namespace ge2e
