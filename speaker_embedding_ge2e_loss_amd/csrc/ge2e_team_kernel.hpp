// GE2E_IMPL_TEAM (ge2e_team.hip): eight workgroups of one XCD share a batch, E is read from HBM once,
// flat 16-row blocks (no padding of a speaker's M rows to an MFMA block).
#pragma once
#include "ge2e_common.hpp"
#include "ge2e_team.hpp"   // formation + hand-off building blocks
#include "ge2e_fused.hpp"  // FusedWs: the layout of the body that redoes a call without teams

namespace ge2e {

// per-team counters, each on a 128-byte line of its own; zeroed by the team_zero_head kernel in front of the launch
struct TeamKFlags {
    unsigned c1;   unsigned pad0[31];   // hand-off 1: unit centroids of a batch published   (+1 per member)
    unsigned c2;   unsigned pad1[31];   // hand-off 2: partial centroid gradients published  (+1 per member)
    unsigned c3;   unsigned pad2[31];   // partial gradients of a batch have been READ       (+1 per speaker)
};

// Per-team exchange area (byte offsets from the team's base): a function of D alone, so the kernel (templated on D)
// folds every offset into an immediate instead of keeping a dozen SGPRs live.
struct TeamKX {
    unsigned chr[2];    // [4 slot tiles][hi, lo][D / 32 K-steps][64 lanes x 16 B]  unit centroids * 2^8 as MFMA fragments (double-buffered)
    unsigned cht[2];    // [8 members][hi, lo][D][8 slots]    the same, one 16-byte k-group per d       (double-buffered)
    unsigned cst[2];    // [64][4] floats                     1/|c|, kappa, |s|, |s|^2                  (double-buffered)
    unsigned sc[2];     // [8][4] floats                      loss, dw, db partials                     (double-buffered)
    unsigned gc;        // [8 members][64 slots][D] floats    partial centroid gradients (single buffer, guarded by c3)
    unsigned stride;    // bytes per team
};
constexpr TeamKX team_exchange(int D) {
    TeamKX x{};
    unsigned o = 0;
    for (int b = 0; b < 2; ++b) { x.chr[b] = o; o += 64u * 2 * D * 2; }
    for (int b = 0; b < 2; ++b) { x.cht[b] = o; o += 8u * 2 * D * 16; }
    for (int b = 0; b < 2; ++b) { x.cst[b] = o; o += 64 * 16; }
    for (int b = 0; b < 2; ++b) { x.sc[b] = o; o += 8 * 16; }
    o = (o + 255u) / 256u * 256u;
    x.gc = o; o += 8u * 64 * D * 4;
    x.stride = (o + 4095u) / 4096u * 4096u;
    return x;
}

struct TeamKWs {
    int spm;            // speaker slots per member = ceil(N / 8) (<= 8)
    int rt;             // rows of a member's images: spm * M rounded up to 16 (<= 80)
    int mul_m;          // ceil(2^16 / M): r / M = (r * mul_m) >> 16 for r < 2^16 / M
    unsigned head_bytes;   // TeamCtl + TeamKFlags[64]
    unsigned xb_bytes, g_bytes;   // LDS regions that are shared by two uses (see the kernel)
    size_t lds_bytes;
    // the in-launch redo (team_finish, ge2e_team.hpp): byte offset of the one-workgroup-per-batch body's workspace slices behind
    // the exchange areas (one slice per workgroup of the team grid), and how many workgroups share the batches of a redo
    size_t fb_off;
    int fb_wgs;
};

int team_fallback_grid(int B);           // workgroups of the gated fall-back launch: one per batch up to one per CU

bool team_supports(int N, int M, int D);
TeamKWs team_layout(int N, int M, int D);
int team_grid(int B);
size_t team_workspace_bytes(int B, int N, int M, int D);
hipError_t launch_team(const Problem& p, hipStream_t stream);
// forward-only calls (p.dE == NULL): the team launch alone, ge2e_team_fwd.hip; launch_team queues the gated fall-back behind it
hipError_t launch_team_fwd(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream);
// LDS of a team launch: the team kernel's own, or the redo body's if that is larger
size_t team_fwd_lds_bytes(int rt, int D);

}  // namespace ge2e
