// extern "C" boundary of libge2e_hip.so (include/ge2e_hip.h): argument checks,
// implementation choice, workspace sizing, launches.  No allocation, no sync, no state.
#include "../../include/ge2e_hip.h"

#include <math.h>
#include <stdlib.h>

#include "ge2e_common.hpp"
#include "ge2e_generic.hpp"
#include "ge2e_helpers.hpp"
#include "ge2e_fused.hpp"
#include "ge2e_selftest.hpp"
#include "ge2e_tiled.hpp"
#include "ge2e_team_kernel.hpp"
#include "ge2e_tail.hpp"
#include "ge2e_wave.hpp"

using namespace ge2e;

namespace {

#ifdef GE2E_PROFILE
unsigned long long* g_prof = nullptr;  // diagnostic build only
#endif

bool shape_ok(int B, int N, int M, int D) { return B >= 1 && N >= 1 && M >= 2 && D >= 1; }

// AUTO at shapes both accept (measured at N=64, M=10, D=256, interleaved in one process, tools/compare_impls.py,
// profiles/r02_team_vs_fused_split_by_B.txt): the eight-CU team kernel wins everywhere (28 us vs 117 us at B = 1,
// 96 vs 133 us at B = 128, 245 vs 298 us at B = 384, 612 vs 619 us at B = 1024, 2.32 vs 2.40 ms at B = 4096, and it moves
// 1.4x instead of 2.5x the algorithmic bytes) except where the one-workgroup-per-batch kernel just fills the chip
// once: B = 256, 166 vs 173 us.
constexpr int kSplitMinB = 208, kSplitMaxB = 256;

// The team kernel wants its workgroups co-resident.  It survives a busy device (waits bounded to milliseconds, then the
// workgroups of the same launch redo the call without teams), but a caller that KNOWS it shares the GPU -- overlapped collectives, several
// processes -- asks for GE2E_IMPL_AUTO_NO_TEAM (the only switch: the GE2E_AUTO_NO_TEAM environment override of rounds 2-3
// is gone, an implementation choice is an argument of the call).

int resolve(int B, int N, int M, int D, int variant, int impl) {
    (void)variant;
    switch (impl) {
        case GE2E_IMPL_AUTO_NO_TEAM:
        case GE2E_IMPL_AUTO:  // split-fp16 MFMA is fp32-grade (tests hold it to 2e-5) and the fastest
            // a few dozen rows: one wave per batch, exact fp32 (31..64 rows: from a few hundred batches per launch on --
            // 117-277 us against 262-408 us at B = 4096, but 30-44 us against 20-28 us for a single batch)
            if (wave_supports(N, M, D) && (!wave_is_large(N, M) || B >= 384)) return GE2E_IMPL_WAVE;
            if (impl == GE2E_IMPL_AUTO && !(B >= kSplitMinB && B <= kSplitMaxB) && N >= 16 && team_supports(N, M, D)) return GE2E_IMPL_TEAM;
            if (fused_split_supports(N, M, D)) return GE2E_IMPL_FUSED_SPLIT;
            return tiled_supports(N, M, D) ? GE2E_IMPL_TILED : GE2E_IMPL_GENERIC;
        case GE2E_IMPL_GENERIC: return GE2E_IMPL_GENERIC;
        case GE2E_IMPL_FUSED_F32: return fused_f32_supports(N, M, D) ? GE2E_IMPL_FUSED_F32 : GE2E_ERR_IMPL;
        case GE2E_IMPL_FUSED_SPLIT: return fused_split_supports(N, M, D) ? GE2E_IMPL_FUSED_SPLIT : GE2E_ERR_IMPL;
        case GE2E_IMPL_TILED: return tiled_supports(N, M, D) ? GE2E_IMPL_TILED : GE2E_ERR_IMPL;
        case GE2E_IMPL_TEAM: return team_supports(N, M, D) ? GE2E_IMPL_TEAM : GE2E_ERR_IMPL;
        case GE2E_IMPL_WAVE: return wave_supports(N, M, D) ? GE2E_IMPL_WAVE : GE2E_ERR_IMPL;
        default: return GE2E_ERR_IMPL;
    }
}

// The first team_head_bytes() of EVERY loss workspace are the team kernel's control block (ge2e_team.hpp), whichever
// implementation runs: the others put their scratch behind it.  A workspace that callers reuse across shapes and
// implementations (the Python module path caches one per stream) therefore keeps a clean block, and a team call is never
// pushed onto its fall-back -- with that kernel's last-bit-different results -- by what ran on the workspace before.
size_t ws_bytes(int B, int N, int M, int D, int impl) {
    size_t own = 0;
    switch (impl) {
        case GE2E_IMPL_GENERIC: own = generic_workspace_bytes(B, N, M, D); break;
        case GE2E_IMPL_FUSED_F32: own = fused_f32_workspace_bytes(B, N, M, D); break;
        case GE2E_IMPL_FUSED_SPLIT: own = fused_split_workspace_bytes(B, N, M, D); break;
        case GE2E_IMPL_TILED: own = tiled_workspace_bytes(B, N, M, D); break;
        case GE2E_IMPL_TEAM: return team_workspace_bytes(B, N, M, D);      // its layout starts with the block
        case GE2E_IMPL_WAVE: return 0;
        default: return 0;
    }
    return own > 0 ? own + team_head_bytes() : 0;
}

int run(Problem& p, int impl, void* workspace, size_t workspace_bytes, void* stream) {
    if (!shape_ok(p.B, p.N, p.M, p.D)) return GE2E_ERR_SHAPE;
    if (p.variant != GE2E_VARIANT_SOFTMAX && p.variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    const int chosen = resolve(p.B, p.N, p.M, p.D, p.variant, impl);
    if (chosen < 0) return chosen;
    const size_t need = ws_bytes(p.B, p.N, p.M, p.D, chosen);
    if (need > 0 && (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255))) return GE2E_ERR_WORKSPACE;
    if (((uintptr_t)p.E & 15) || ((uintptr_t)p.dE & 15)) return GE2E_ERR_ALIGN;
    p.ws = (float*)((char*)workspace + (need > 0 && chosen != GE2E_IMPL_TEAM ? team_head_bytes() : 0));
#ifdef GE2E_PROFILE
    p.prof = g_prof;
#endif
    p.log_eps = p.eps > 0.f ? logf(p.eps) : -INFINITY;
    hipError_t err = hipSuccess;
    switch (chosen) {
        case GE2E_IMPL_GENERIC: err = launch_generic(p, (hipStream_t)stream); break;
        case GE2E_IMPL_FUSED_F32: err = launch_fused_f32(p, (hipStream_t)stream); break;
        case GE2E_IMPL_FUSED_SPLIT: err = launch_fused_split(p, (hipStream_t)stream); break;
        case GE2E_IMPL_TILED: err = launch_tiled(p, (hipStream_t)stream); break;
        case GE2E_IMPL_TEAM: err = launch_team(p, (hipStream_t)stream); break;
        case GE2E_IMPL_WAVE: err = launch_wave(p, (hipStream_t)stream); break;
        default: return GE2E_ERR_IMPL;
    }
    return (int)err;
}

}  // namespace

extern "C" {

#ifdef GE2E_PROFILE
// diagnostic build only (tools/profile_phases.py): where the kernels add their phase stamps
void ge2e_debug_set_prof(void* device_u64_buffer) { g_prof = (unsigned long long*)device_u64_buffer; }
#endif

int ge2e_abi_version(void) { return GE2E_ABI_VERSION; }

const char* ge2e_strerror(int code) {
    switch (code) {
        case GE2E_OK: return "ok";
        case GE2E_ERR_NULL: return "required pointer is NULL";
        case GE2E_ERR_SHAPE: return "bad shape: need B,N,D >= 1 and M >= 2";
        case GE2E_ERR_WORKSPACE: return "workspace missing, too small or not 256-byte aligned";
        case GE2E_ERR_VARIANT: return "unknown loss variant";
        case GE2E_ERR_IMPL: return "requested implementation cannot run this shape";
        case GE2E_ERR_ALIGN: return "E / dE must be 16-byte aligned";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int ge2e_resolve_impl(int B, int N, int M, int D, int variant, int impl) {
    if (!shape_ok(B, N, M, D)) return GE2E_ERR_SHAPE;
    return resolve(B, N, M, D, variant, impl);
}

size_t ge2e_workspace_bytes(int B, int N, int M, int D, int variant, int impl) {
    if (!shape_ok(B, N, M, D)) return 0;
    const int chosen = resolve(B, N, M, D, variant, impl);
    return chosen < 0 ? 0 : ws_bytes(B, N, M, D, chosen);
}

int ge2e_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    if (!workspace) return GE2E_ERR_NULL;
    if (workspace_bytes < team_head_bytes()) return GE2E_OK;      // too small for any team launch: nothing to prepare
    return (int)launch_team_head_init(workspace, team_head_bytes(), false, (hipStream_t)stream);   // > 0: a hipError_t, like every launch
}

int ge2e_loss_fwd_bwd(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                      float eps_cos, float eps, int variant, int impl, float* loss,
                      float* per_emb_loss, float* dE, float* dw, float* db, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!E || !w || !b || !loss) return GE2E_ERR_NULL;
    if (dE && (!dw || !db)) return GE2E_ERR_NULL;
    Problem p{};
    p.E = E; p.w = w; p.b = b; p.loss = loss; p.per = per_emb_loss;
    p.dE = dE; p.dw = dw; p.db = db; p.cos_out = nullptr;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = variant; p.eps_cos = eps_cos; p.eps = eps;
    return run(p, impl, workspace, workspace_bytes, stream);
}

int ge2e_raw_supported(int N, int M, int D) { return (N >= 1 && M >= 2 && D >= 1 && wave_supports_raw(N, M, D)) ? 1 : 0; }

int ge2e_loss_fwd_bwd_raw(const float* Y, const int* src, int B, int N, int M, int D, const float* w, const float* b,
                          float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dY, float* dw,
                          float* db, void* stream) {
    if (!Y || !w || !b || !loss) return GE2E_ERR_NULL;
    if (dY && (!dw || !db)) return GE2E_ERR_NULL;
    if (!shape_ok(B, N, M, D)) return GE2E_ERR_SHAPE;
    if (variant != GE2E_VARIANT_SOFTMAX && variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    if (!wave_supports_raw(N, M, D)) return GE2E_ERR_IMPL;
    if (((uintptr_t)Y & 15) || ((uintptr_t)dY & 15)) return GE2E_ERR_ALIGN;
    Problem p{};
    p.E = Y; p.w = w; p.b = b; p.loss = loss; p.per = per_emb_loss;
    p.dE = dY; p.dw = dw; p.db = db; p.cos_out = nullptr;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = variant; p.eps_cos = eps_cos; p.eps = eps;
    p.log_eps = eps > 0.f ? logf(eps) : -INFINITY;
    p.raw = 1; p.src = src;
    return (int)launch_wave(p, (hipStream_t)stream);
}

// Forward-only similarity matrix; w and b are not applied (s5:44 applies its own).
// get_cos_sim runs on the matrix cores (TILED's preparation pass + split-fp16 similarity contraction + a row pass for the
// leave-one-out column) where that kernel takes the shape and the contraction is big enough to pay for three launches;
// tiny shapes (the reference's N = 2..6) stay on the exact-fp32 VALU kernel.
static bool cos_on_mfma(int N, int M, int D) { return N >= 16 && tiled_supports(N, M, D); }

size_t ge2e_cos_sim_workspace_bytes(int B, int N, int M, int D) {
    if (!shape_ok(B, N, M, D)) return 0;
    const size_t g = ws_bytes(B, N, M, D, GE2E_IMPL_GENERIC);      // what run() asks for (the control block's room included)
    const size_t t = cos_on_mfma(N, M, D) ? tiled_workspace_bytes(B, N, M, D) : 0;
    return g > t ? g : t;
}

int ge2e_cos_sim(const float* E, int B, int N, int M, int D, float eps_cos, float eps, float* cos,
                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!E || !cos) return GE2E_ERR_NULL;
    Problem p{};
    p.E = E; p.w = nullptr; p.b = nullptr; p.w_imm = 1.0f; p.b_imm = 0.0f; p.cos_out = cos;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = GE2E_VARIANT_SOFTMAX; p.eps_cos = eps_cos; p.eps = eps;
    // (a caller that sized the workspace for the VALU kernel only -- ge2e_workspace_bytes(.., GE2E_IMPL_GENERIC), ABI 1's
    // rule -- gets the VALU kernel)
    if (shape_ok(B, N, M, D) && cos_on_mfma(N, M, D) && workspace && workspace_bytes >= tiled_workspace_bytes(B, N, M, D) &&
        !((uintptr_t)workspace & 255) && !((uintptr_t)E & 15)) {
        p.ws = (float*)workspace;
        p.log_eps = eps > 0.f ? logf(eps) : -INFINITY;
        return (int)launch_tiled_cos(p, (hipStream_t)stream);
    }
    return run(p, GE2E_IMPL_GENERIC, workspace, workspace_bytes, stream);
}

// get_cos_sim with the caller's centroids (the reference's second argument), forward only.
int ge2e_cos_sim_centroids(const float* E, const float* C, int B, int N, int M, int D, float eps_cos, float eps,
                           float* cos, void* stream) {
    if (!E || !C || !cos) return GE2E_ERR_NULL;
    if (!shape_ok(B, N, M, D)) return GE2E_ERR_SHAPE;
    return (int)launch_cos_centroids(E, C, B, N, N, 0, M, D, eps_cos, eps, cos, (hipStream_t)stream);
}

int ge2e_calc_loss(const float* sim, int B, int N, int M, float eps, int variant, float* loss,
                   float* per_emb_loss, void* stream) {
    if (!sim || !loss) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1) return GE2E_ERR_SHAPE;
    if (variant != GE2E_VARIANT_SOFTMAX && variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    return (int)launch_calc_loss(sim, B, N, N, 0, M, eps, variant, loss, per_emb_loss, (hipStream_t)stream);
}

int ge2e_centroids(const float* E, int B, int N, int M, int D, float* cent, void* stream) {
    if (!E || !cent) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1 || D < 1) return GE2E_ERR_SHAPE;
    return (int)launch_centroids(E, B, N, M, D, cent, (hipStream_t)stream);
}

int ge2e_utterance_centroids(const float* E, int B, int N, int M, int D, float* U, void* stream) {
    if (!E || !U) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 2 || D < 1) return GE2E_ERR_SHAPE;
    return (int)launch_utt_centroids(E, B, N, M, D, U, (hipStream_t)stream);
}

int ge2e_centroids_bwd(const float* g_cent, int B, int N, int M, int D, float* dE, void* stream) {
    if (!g_cent || !dE) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1 || D < 1) return GE2E_ERR_SHAPE;
    return (int)launch_centroids_bwd(g_cent, B, N, M, D, dE, (hipStream_t)stream);
}

size_t ge2e_cos_sim_bwd_workspace_bytes(int B, int N, int M, int D) {
    return shape_ok(B, N, M, D) ? cos_bwd_workspace_bytes(B, N, N, M, D) : 0;
}

int ge2e_cos_sim_bwd(const float* E, const float* C, const float* cos, const float* g_cos, int B, int N, int M, int D,
                     float eps_cos, float eps, float* dE, float* dC, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (!E || !C || !cos || !g_cos || !dE || !dC) return GE2E_ERR_NULL;
    if (!shape_ok(B, N, M, D)) return GE2E_ERR_SHAPE;
    if (!workspace || workspace_bytes < cos_bwd_workspace_bytes(B, N, N, M, D) || ((uintptr_t)workspace & 15)) return GE2E_ERR_WORKSPACE;
    return (int)launch_cos_bwd(E, C, cos, g_cos, B, N, N, 0, M, D, eps_cos, eps, dE, dC, (float*)workspace, (hipStream_t)stream);
}

int ge2e_calc_loss_bwd(const float* sim, int B, int N, int M, float eps, int variant, const float* g_loss,
                       const float* g_per, float* d_sim, void* stream) {
    if (!sim || !d_sim || (!g_loss && !g_per)) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1) return GE2E_ERR_SHAPE;
    if (variant != GE2E_VARIANT_SOFTMAX && variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    return (int)launch_calc_loss_bwd(sim, B, N, N, 0, M, eps, variant, g_loss, g_per, d_sim, (hipStream_t)stream);
}

// ---- the same helpers on LOCAL ROWS (the speaker-sharded loss, s3:42-80 / s3:114-127 semantics) --------------
static bool rows_ok(int B, int n, int N, int j0, int M, int D) {
    return B >= 1 && n >= 1 && N >= 1 && M >= 2 && D >= 1 && j0 >= 0 && j0 + n <= N;
}
int ge2e_cos_sim_rows(const float* E, const float* C, int B, int n, int N, int j0, int M, int D, float eps_cos, float eps,
                      float* cos, void* stream) {
    if (!E || !C || !cos) return GE2E_ERR_NULL;
    if (!rows_ok(B, n, N, j0, M, D)) return GE2E_ERR_SHAPE;
    return (int)launch_cos_centroids(E, C, B, n, N, j0, M, D, eps_cos, eps, cos, (hipStream_t)stream);
}
size_t ge2e_cos_sim_rows_bwd_workspace_bytes(int B, int n, int N, int M, int D) {
    return rows_ok(B, n, N, 0, M, D) ? cos_bwd_workspace_bytes(B, n, N, M, D) : 0;
}
int ge2e_cos_sim_rows_bwd(const float* E, const float* C, const float* cos, const float* g_cos, int B, int n, int N, int j0,
                          int M, int D, float eps_cos, float eps, float* dE, float* dC, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (!E || !C || !cos || !g_cos || !dE || !dC) return GE2E_ERR_NULL;
    if (!rows_ok(B, n, N, j0, M, D)) return GE2E_ERR_SHAPE;
    if (!workspace || workspace_bytes < cos_bwd_workspace_bytes(B, n, N, M, D) || ((uintptr_t)workspace & 15)) return GE2E_ERR_WORKSPACE;
    return (int)launch_cos_bwd(E, C, cos, g_cos, B, n, N, j0, M, D, eps_cos, eps, dE, dC, (float*)workspace, (hipStream_t)stream);
}
int ge2e_calc_loss_rows(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, float* loss,
                        float* per_emb_loss, void* stream) {
    if (!sim || !loss) return GE2E_ERR_NULL;
    if (!rows_ok(B, n, N, j0, M, 1)) return GE2E_ERR_SHAPE;
    if (variant != GE2E_VARIANT_SOFTMAX && variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    return (int)launch_calc_loss(sim, B, n, N, j0, M, eps, variant, loss, per_emb_loss, (hipStream_t)stream);
}
int ge2e_calc_loss_rows_bwd(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, const float* g_loss,
                            const float* g_per, float* d_sim, void* stream) {
    if (!sim || !d_sim || (!g_loss && !g_per)) return GE2E_ERR_NULL;
    if (!rows_ok(B, n, N, j0, M, 1)) return GE2E_ERR_SHAPE;
    if (variant != GE2E_VARIANT_SOFTMAX && variant != GE2E_VARIANT_CONTRAST) return GE2E_ERR_VARIANT;
    return (int)launch_calc_loss_bwd(sim, B, n, N, j0, M, eps, variant, g_loss, g_per, d_sim, (hipStream_t)stream);
}

int ge2e_scale_grads(const float* dE, const float* dw, const float* db, const float* g, int g_count, int B, int N, int M,
                     int D, float* gE, float* gw, float* gb, void* stream) {
    if (!g || (gE && !dE) || (gw && !dw) || (gb && !db) || (!gE && !gw && !gb)) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1 || D < 1 || (g_count != 1 && g_count != B)) return GE2E_ERR_SHAPE;
    return (int)launch_scale_grads(dE, dw, db, g, g_count, B, (size_t)N * M * D, gE, gw, gb, (hipStream_t)stream);
}

int ge2e_normalize_unperm(const float* y, const int* src, int rows, int D, float* e, float* rnorm, void* stream) {
    if (!y || !e || !rnorm) return GE2E_ERR_NULL;
    if (rows < 1 || D < 1) return GE2E_ERR_SHAPE;
    return (int)launch_tail_fwd(y, src, rows, D, e, rnorm, (hipStream_t)stream);
}

int ge2e_normalize_unperm_bwd(const float* g, const float* e, const float* rnorm, const int* src, int rows, int D,
                              float* dy, void* stream) {
    if (!g || !e || !rnorm || !dy) return GE2E_ERR_NULL;
    if (rows < 1 || D < 1) return GE2E_ERR_SHAPE;
    return (int)launch_tail_bwd(g, e, rnorm, src, rows, D, dy, (hipStream_t)stream);
}

int ge2e_eer_counts(const float* sim, int B, int N, int M, const float* thresholds, int T, int* counts, void* stream) {
    if (!sim || !thresholds || !counts) return GE2E_ERR_NULL;
    if (B < 1 || N < 1 || M < 1 || T < 1 || T > 4096) return GE2E_ERR_SHAPE;
    return (int)launch_eer_counts(sim, B, N, M, thresholds, T, counts, (hipStream_t)stream);
}

int ge2e_sample_batch(const void* store, int store_is_f64, const long long* spk_offsets, const int* utter_idx,
                      const int* clip_start, int N, int M, int T, int L, int F, float* out, void* stream) {
    if (!store || !spk_offsets || !utter_idx || !clip_start || !out) return GE2E_ERR_NULL;
    if (N < 1 || M < 1 || T < 1 || L < 1 || L > T || F < 1 || (long long)N * M > 65535) return GE2E_ERR_SHAPE;
    return (int)launch_sample_batch(store, store_is_f64, spk_offsets, utter_idx, clip_start, N, M, T, L, F, out,
                                    (hipStream_t)stream);
}

// GE2E_IMPL_TEAM with its abort word raised before the launch: no team forms, the launch redoes the call with one workgroup per batch.
int ge2e_selftest_team_fallback(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                                float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dE,
                                float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (!E || !w || !b || !loss) return GE2E_ERR_NULL;
    if (dE && (!dw || !db)) return GE2E_ERR_NULL;
    Problem p{};
    p.E = E; p.w = w; p.b = b; p.loss = loss; p.per = per_emb_loss;
    p.dE = dE; p.dw = dw; p.db = db; p.cos_out = nullptr;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = variant; p.eps_cos = eps_cos; p.eps = eps;
    p.test_abort = 1;
    return run(p, GE2E_IMPL_TEAM, workspace, workspace_bytes, stream);
}

// GE2E_IMPL_TEAM with the abort word raised by one workgroup in the middle of the grid's finish (some leave, some stay)
int ge2e_selftest_team_abort_midgrid(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                                     float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dE,
                                     float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    if (!E || !w || !b || !loss) return GE2E_ERR_NULL;
    if (dE && (!dw || !db)) return GE2E_ERR_NULL;
    Problem p{};
    p.E = E; p.w = w; p.b = b; p.loss = loss; p.per = per_emb_loss;
    p.dE = dE; p.dw = dw; p.db = db; p.cos_out = nullptr;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = variant; p.eps_cos = eps_cos; p.eps = eps;
    p.test_abort = 2;
    return run(p, GE2E_IMPL_TEAM, workspace, workspace_bytes, stream);
}

// GE2E_IMPL_TEAM on at most `max_workgroups` workgroups (a multiple of 64: eight per XCD form one team): many batches
// through few teams, so that the hand-off counters of a team run far beyond what a full-size launch reaches.
int ge2e_selftest_team_grid(const float* E, int B, int N, int M, int D, const float* w, const float* b, float eps_cos,
                            float eps, int variant, float* loss, float* per_emb_loss, float* dE, float* dw, float* db,
                            void* workspace, size_t workspace_bytes, void* stream, int max_workgroups) {
    if (!E || !w || !b || !loss) return GE2E_ERR_NULL;
    if (dE && (!dw || !db)) return GE2E_ERR_NULL;
    if (max_workgroups < 64) return GE2E_ERR_SHAPE;
    Problem p{};
    p.E = E; p.w = w; p.b = b; p.loss = loss; p.per = per_emb_loss;
    p.dE = dE; p.dw = dw; p.db = db; p.cos_out = nullptr;
    p.B = B; p.N = N; p.M = M; p.D = D; p.variant = variant; p.eps_cos = eps_cos; p.eps = eps;
    p.grid_cap = max_workgroups;
    return run(p, GE2E_IMPL_TEAM, workspace, workspace_bytes, stream);
}

int ge2e_selftest_split_gemm(const float* A, const float* Bm, const float* G, float* X, float* GE, float* GC,
                             void* stream) {
    if (!A || !Bm || !G || !X || !GE || !GC) return GE2E_ERR_NULL;
    return (int)launch_selftest_split(A, Bm, G, X, GE, GC, (hipStream_t)stream);
}

int ge2e_selftest_wave_ops(const float* x, float* out, void* stream) {
    if (!x || !out) return GE2E_ERR_NULL;
    return (int)launch_selftest_wave(x, out, (hipStream_t)stream);
}

int ge2e_selftest_rows16(const float* CH, const float* R, float* XT, float* GE, float* GT, void* stream) {
    if (!CH || !R || !XT || !GE || !GT) return GE2E_ERR_NULL;
    return (int)launch_selftest_rows16(CH, R, XT, GE, GT, (hipStream_t)stream);
}

size_t ge2e_selftest_team_bytes(int payload_f4) { return payload_f4 > 0 ? selftest_team_bytes(payload_f4) : 0; }

int ge2e_selftest_team(void* ws, size_t ws_bytes, int grid, int rounds, int payload_f4, unsigned* out, void* stream) {
    if (!ws || !out) return GE2E_ERR_NULL;
    if (rounds < 1 || payload_f4 < 1) return GE2E_ERR_SHAPE;
    return (int)launch_selftest_team(ws, ws_bytes, grid, rounds, payload_f4, out, (hipStream_t)stream);
}

}  // extern "C"
