// GE2E_IMPL_TEAM2 (ge2e_team2.hip): eight workgroups of one XCD share a batch, E is read from HBM once,
// flat 16-row blocks (no padding of a speaker's M rows to an MFMA block).
#pragma once
#include "ge2e_common.hpp"
#include "ge2e_team.hpp"

namespace ge2e {

// per-team counters, each on a 128-byte line of its own; zeroed by the memset node in front of the launch
struct Team2Flags {
    unsigned c1;   unsigned pad0[31];   // hand-off 1: unit centroids of a batch published   (+1 per member)
    unsigned c2;   unsigned pad1[31];   // hand-off 2: partial centroid gradients published  (+1 per member)
    unsigned c3;   unsigned pad2[31];   // partial gradients of a batch have been READ       (+1 per speaker)
};

struct Team2Ws {
    int spm;            // speaker slots per member = ceil(N / 8) (<= 8)
    int rt;             // rows of a member's images: spm * M rounded up to 16 (<= 80)
    int mul_m;          // ceil(2^16 / M): r / M = (r * mul_m) >> 16 for r < 2^16 / M
    // per-team exchange area, byte offsets from the team's base
    unsigned chr[2];    // [64 slots][hi D | lo D] halfs      unit centroids * 2^8, row-major          (double-buffered)
    unsigned cht[2];    // [8 members][hi, lo][D][8 slots]    the same, one 16-byte k-group per d       (double-buffered)
    unsigned cst[2];    // [64][4] floats                     1/|c|, kappa, |s|, |s|^2                  (double-buffered)
    unsigned sc[2];     // [8][4] floats                      loss, dw, db partials                     (double-buffered)
    unsigned gc;        // [8 members][64 slots][D] floats    partial centroid gradients (single buffer, guarded by c3)
    size_t stride;      // bytes per team
    size_t head_bytes;  // TeamCtl + Team2Flags[64]
    size_t fb_off;      // workspace of the gated fall-back launch (bytes from the workspace base)
    size_t lds_bytes;
    unsigned xb_bytes, g_bytes;   // LDS regions that are shared by two uses (see the kernel)
};

constexpr int TEAM2_FALLBACK_GRID = 32;   // workgroups of the gated fall-back launch

bool team2_supports(int N, int M, int D);
Team2Ws team2_layout(int N, int M, int D);
int team2_grid(int B);
size_t team2_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_team2(const Problem& p, hipStream_t stream);

}  // namespace ge2e
