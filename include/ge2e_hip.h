/*
 * ge2e_hip.h -- C ABI of libge2e_hip.so: the GE2E speaker-verification loss
 * (forward + full gradient) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * This is the drop-in boundary for the hot path of
 * gkv856/speaker_embedding_GE2E_loss:
 *     embedding_model_GE2E/s3_loss_function_GE2E.py:19-30   GE2ELoss.forward
 *     embedding_model_GE2E/s3_loss_function_GE2E.py:34-127  get_centroids,
 *         get_cos_sim, get_utterance_centroids, calc_loss
 *     embedding_model_GE2E/s4_train_embed_model.py:200      loss.backward()
 * The reference has no native layer; these entry points are what a ctypes /
 * torch binding for that path binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless it says host;
 *   - all tensors are dense, row-major, float32;  E is [B][N][M][D]
 *     (B independent (speakers x utterances) batches, N speakers, M utterances
 *     per speaker, D embedding size); B = 1 is the reference's single batch;
 *   - the caller owns every buffer including the workspace; the library
 *     allocates nothing, keeps no state and never synchronises: calls only
 *     enqueue work on `stream` (a hipStream_t passed as void*, NULL = default);
 *   - a loss workspace (ge2e_workspace_bytes) begins with a 26 KB control block
 *     of the eight-CU team kernel, whichever implementation runs on it; the
 *     block cleans itself after every call (see ge2e_workspace_init);
 *   - w and b (s3:16-17) are read from device memory, no host sync;
 *   - return value: 0 = ok, < 0 = argument error (GE2E_ERR_*), > 0 = hipError_t
 *     of a failed launch.  Nothing is thrown across the boundary.
 */
#ifndef GE2E_HIP_H
#define GE2E_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GE2E_ABI_VERSION 2   /* 2: ge2e_workspace_init, the *_rows helpers; every loss workspace starts with the control block */

/* loss variants: eq. (6) softmax is the reference's (s3:115-127); eq. (7)
 * contrast is defined from arXiv:1710.10467 (absent from the reference). */
#define GE2E_VARIANT_SOFTMAX 0
#define GE2E_VARIANT_CONTRAST 1

/* kernel selection */
#define GE2E_IMPL_AUTO 0        /* fastest implementation valid for the shape   */
#define GE2E_IMPL_GENERIC 1     /* one workgroup per batch, fp32 VALU, any shape */
#define GE2E_IMPL_FUSED_F32 2   /* one workgroup per batch, LDS-resident centroids,
                                   exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)      */
#define GE2E_IMPL_FUSED_SPLIT 3 /* as FUSED_F32 with fp16 hi+lo split operands on
                                   v_mfma_f32_32x32x16_f16, fp32 accumulate      */
#define GE2E_IMPL_TILED 4       /* many workgroups per batch (large N / D, small B); D any multiple of 8 up to 1024 */
#define GE2E_IMPL_TEAM 5        /* eight workgroups of one XCD per batch, the member's rows resident in LDS: E is
                                   read once.  ONE launch per call: if the teams cannot form or a hand-off times
                                   out, the same workgroups redo the call with FUSED_SPLIT's one-workgroup-per-batch
                                   body before the launch ends (see GE2E_IMPL_AUTO_NO_TEAM).  D: any multiple of 4
                                   up to 256 (padded to the next multiple of 64 inside the kernel)               */

#define GE2E_IMPL_WAVE 6        /* one WAVE per batch, the batch in registers, exact fp32, no workspace: the reference's
                                   own shapes (up to 64 rows: N <= 3..12 depending on M in {2,3,4,5,6,8,10,16},
                                   D <= 256, D % 4 == 0)                                                        */
#define GE2E_IMPL_AUTO_NO_TEAM 7 /* AUTO without GE2E_IMPL_TEAM: for callers that KNOW other streams or processes keep
                                   CUs busy while the loss runs (overlapped collectives, a shared GPU).  The team kernel
                                   wants its workgroups co-resident; it survives a busy device (waits bounded to a few
                                   milliseconds, then the in-launch redo), but not choosing it saves those waits.  */

#define GE2E_OK 0
#define GE2E_ERR_NULL (-1)      /* a required pointer is NULL                    */
#define GE2E_ERR_SHAPE (-2)     /* B,N,D < 1 or M < 2 (M = 1 divides by zero in
                                   the reference, s3:110-111)                    */
#define GE2E_ERR_WORKSPACE (-3) /* workspace NULL/too small/misaligned (256 B)   */
#define GE2E_ERR_VARIANT (-4)
#define GE2E_ERR_IMPL (-5)      /* requested impl cannot run this shape          */
#define GE2E_ERR_ALIGN (-6)     /* E / dE not 16-byte aligned                    */

int ge2e_abi_version(void);
const char* ge2e_strerror(int code);

/* Which implementation GE2E_IMPL_AUTO resolves to for a shape (or `impl`
 * itself if it is valid for the shape, GE2E_ERR_IMPL if not). */
int ge2e_resolve_impl(int B, int N, int M, int D, int variant, int impl);

/* Bytes of scratch ge2e_loss_fwd_bwd / ge2e_cos_sim need for this shape. */
size_t ge2e_workspace_bytes(int B, int N, int M, int D, int variant, int impl);

/* OPTIONAL, once per workspace allocation (enqueue-only, one 2-us launch): writes the clean control block GE2E_IMPL_TEAM
 * expects at the head of its workspace.  The block is self-cleaning -- every call hands it back the way it found it, so
 * the steady state has no zeroing launch in front of the kernel -- and a workspace that was NOT initialised (or that another
 * implementation has used in between) is still safe: the first call on it is computed without teams (one workgroup
 * per batch, inside the same launch), which leaves a clean block behind.  Calling this just makes the first call already run the team kernel.  Workspaces
 * smaller than the block (26 KB, the first bytes of every loss workspace whichever implementation runs) are left alone.  (The reference has no counterpart: s3's module allocates nothing.) */
int ge2e_workspace_init(void* workspace, size_t workspace_bytes, void* stream);

/*
 * GE2ELoss.forward + its autograd backward in one call (s3:19-30 + s4:200).
 *   loss          [B]        sum over the (N,M) per-utterance losses (s3:126)
 *   per_emb_loss  [B][N][M]  calc_loss()'s second return value, or NULL
 *   dE            [B][N][M][D] dLoss/dE, or NULL for forward only
 *   dw, db        [B]        dLoss/dw, dLoss/db (NULL allowed when dE is NULL)
 * eps_cos = F.cosine_similarity's eps (1e-8); eps = hp.general.small_err (1e-6).
 */
int ge2e_loss_fwd_bwd(const float* E, int B, int N, int M, int D,
                      const float* w, const float* b, float eps_cos, float eps,
                      int variant, int impl,
                      float* loss, float* per_emb_loss,
                      float* dE, float* dw, float* db,
                      void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same, fed with the encoder's RAW output (s2_model_GE2E_loss_speach_embed.py:34 +
 * s4_train_embed_model.py:186-192 folded into the loss kernel's load and store stages):
 *   Y   [B][N*M][D]  the encoder's projection BEFORE its L2-normalisation, rows in the encoder's own (permuted) order
 *   src [B][N*M] int32 or NULL: row r of the (N,M,D) block is Y[src[r]] / |Y[src[r]]|  (the reference's `unperm`; NULL =
 *       identity).  Each batch's src must be a permutation of 0..N*M-1.
 *   dY  [B][N*M][D]  dLoss/dY, through the normalisation's backward (g - e (e . g)) / |y|, in Y's row order; or NULL.
 * One launch instead of normalise + gather, loss, and the normalisation's backward + scatter; no (N,M,D) intermediate.
 * Shapes: ge2e_raw_supported(N, M, D) != 0 (the one-wave-per-batch kernel's register-only shapes: the reference's training
 * shapes N=2 M=16, N=4 M=5, ...); otherwise GE2E_ERR_IMPL and the caller uses ge2e_normalize_unperm + ge2e_loss_fwd_bwd.
 */
int ge2e_raw_supported(int N, int M, int D);
int ge2e_loss_fwd_bwd_raw(const float* Y, const int* src, int B, int N, int M, int D,
                          const float* w, const float* b, float eps_cos, float eps, int variant,
                          float* loss, float* per_emb_loss, float* dY, float* dw, float* db, void* stream);

/* GE2ELoss.get_cos_sim (s3:42-80): cos [B][N][M][N], leave-one-out centroid on
 * the own-speaker column, eps added to every entry.  Forward-only consumer:
 * s5_eval_model.py:42-44.
 * Workspace: ge2e_cos_sim_workspace_bytes(B, N, M, D).  With it, shapes the tiled kernel takes (N >= 16, D % 64 == 0) run
 * the N x (N M) x D contraction on the matrix cores (split-fp16 x3, fp32 accumulate: cosines to ~1e-6); with the smaller
 * ge2e_workspace_bytes(.., GE2E_IMPL_GENERIC) of ABI 1's rule, or any other shape, the exact-fp32 VALU kernel runs. */
size_t ge2e_cos_sim_workspace_bytes(int B, int N, int M, int D);
int ge2e_cos_sim(const float* E, int B, int N, int M, int D,
                 float eps_cos, float eps, float* cos,
                 void* workspace, size_t workspace_bytes, void* stream);

/* GE2ELoss.calc_loss (s3:115-127) on a caller-made similarity matrix
 * sim [B][N][M][N]:  loss [B], per_emb_loss [B][N][M] (or NULL).  Forward only. */
int ge2e_calc_loss(const float* sim, int B, int N, int M, float eps, int variant,
                   float* loss, float* per_emb_loss, void* stream);

/* GE2ELoss.get_cos_sim(embeddings, centroids, hp) (s3:42-80) with the CALLER'S centroids C [B][N][D]: other-speaker
 * columns use C, the own-speaker column the leave-one-out centroid of E; + eps on every entry.  cos [B][N][M][N].
 * (ge2e_cos_sim above is the special case C = get_centroids(E), the only one the reference's callers use.) */
int ge2e_cos_sim_centroids(const float* E, const float* C, int B, int N, int M, int D, float eps_cos,
                           float eps, float* cos, void* stream);

/* GE2ELoss.get_centroids (s3:34-38): cent [B][N][D] = mean over M. */
int ge2e_centroids(const float* E, int B, int N, int M, int D, float* cent, void* stream);

/* GE2ELoss.get_utterance_centroids (s3:95-112): U [B][N][M][D], u_ji = (sum_i' e_ji' - e_ji) / (M - 1).  The map is
 * linear and symmetric, so its backward is the same call on the incoming gradient. */
int ge2e_utterance_centroids(const float* E, int B, int N, int M, int D, float* U, void* stream);

/* ---- backward passes of the static helpers (the reference's are plain autograd code: s3:33-38, 41-80, 114-127) ----
 * get_centroids:  dE [B][N][M][D] = g_cent [B][N][D] / M */
int ge2e_centroids_bwd(const float* g_cent, int B, int N, int M, int D, float* dE, void* stream);
/* get_cos_sim(E, C): from the forward result `cos` and its incoming gradient g_cos (both [B][N][M][N]) the gradients
 * with respect to the embeddings, dE [B][N][M][D] (a-slot + leave-one-out slot), and to the caller's centroids,
 * dC [B][N][D].  Scratch: ge2e_cos_sim_bwd_workspace_bytes, 16-byte aligned. */
size_t ge2e_cos_sim_bwd_workspace_bytes(int B, int N, int M, int D);
int ge2e_cos_sim_bwd(const float* E, const float* C, const float* cos, const float* g_cos, int B, int N, int M, int D,
                     float eps_cos, float eps, float* dE, float* dC, void* workspace, size_t workspace_bytes,
                     void* stream);
/* calc_loss(sim): d_sim [B][N][M][N] from g_loss [B] (gradient of the summed loss, or NULL) and g_per [B][N][M]
 * (gradient of per_embedding_loss, or NULL). */
int ge2e_calc_loss_bwd(const float* sim, int B, int N, int M, float eps, int variant, const float* g_loss,
                       const float* g_per, float* d_sim, void* stream);

/*
 * The static helpers on LOCAL ROWS -- the speaker-sharded exact loss: a rank holds the rows of the n
 * speakers j0 .. j0 + n - 1 of a batch of N speakers and has gathered all N centroids.  Same semantics as
 * get_cos_sim(embeddings, centroids) (s3:42-80: the caller's centroids on every other-speaker column, the leave-one-out
 * centroid of the LOCAL rows on the own column j0 + jl) and calc_loss (s3:114-127), restricted to those rows:
 *   E   [B][n][M][D]   C [B][N][D]   cos / sim / g_cos / d_sim [B][n][M][N]   loss [B]   per_emb_loss [B][n][M]
 *   dE  [B][n][M][D]   dC [B][N][D] = this shard's PARTIAL centroid gradient (summed over the shards by the caller)
 * n = N, j0 = 0 is exactly ge2e_cos_sim_centroids / ge2e_cos_sim_bwd / ge2e_calc_loss / ge2e_calc_loss_bwd.
 */
int ge2e_cos_sim_rows(const float* E, const float* C, int B, int n, int N, int j0, int M, int D, float eps_cos, float eps,
                      float* cos, void* stream);
size_t ge2e_cos_sim_rows_bwd_workspace_bytes(int B, int n, int N, int M, int D);
int ge2e_cos_sim_rows_bwd(const float* E, const float* C, const float* cos, const float* g_cos, int B, int n, int N, int j0,
                          int M, int D, float eps_cos, float eps, float* dE, float* dC, void* workspace,
                          size_t workspace_bytes, void* stream);
int ge2e_calc_loss_rows(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, float* loss,
                        float* per_emb_loss, void* stream);
int ge2e_calc_loss_rows_bwd(const float* sim, int B, int n, int N, int j0, int M, float eps, int variant, const float* g_loss,
                            const float* g_per, float* d_sim, void* stream);

/* What `loss.backward()` (s4:200) does with the results of ge2e_loss_fwd_bwd: scale by the incoming gradient g (device;
 * g_count = 1 for a scalar loss or B for a per-batch loss vector) in ONE launch:
 *   gE [B][N][M][D] = g[b] dE[b]   (NULL: skip);   gw [1] = sum_b g[b] dw[b];   gb [1] = sum_b g[b] db[b]   (NULL: skip) */
int ge2e_scale_grads(const float* dE, const float* dw, const float* db, const float* g, int g_count, int B, int N, int M,
                     int D, float* gE, float* gw, float* gb, void* stream);

/* ---- the callers either side of the loss ---------------------------------------------------------
 * Encoder tail: the L2-normalisation that ends the encoder's forward (s2_model_GE2E_loss_speach_embed.py:34), the
 * un-permute gather `embeddings[unperm]` (s4_train_embed_model.py:186) and the (N,M,D) reshape (s4:189) in one pass:
 *   e[i][:] = y[src[i]][:] / |y[src[i]]|,  rnorm[i] = 1 / |y[src[i]]|      y, e [rows][D]; src [rows] int32 or NULL (identity)
 * src must be a permutation of 0..rows-1 (the reference's `unperm`); no epsilon, like s2:34. */
int ge2e_normalize_unperm(const float* y, const int* src, int rows, int D, float* e, float* rnorm, void* stream);
/* its backward: dy[src[i]][:] = (g[i] - e[i] (e[i] . g[i])) * rnorm[i]  (every row of dy is written exactly once). */
int ge2e_normalize_unperm_bwd(const float* g, const float* e, const float* rnorm, const int* src, int rows, int D,
                              float* dy, void* stream);
/* Threshold sweep of calculate_ERR (s5_eval_model.py:57-98) on a similarity matrix sim [B][N][M][N]: for every threshold
 * (non-decreasing fp32 table, T <= 4096; the reference's is 0.5 + 0.01 i, i < 50, s5:57) the two integer counts the
 * reference derives FAR and FRR from, counts [B][T][2] int32:
 *   [0] = #{(j,i,k), k != j : sim[j][i][k] > thr}   (s5:82)      [1] = #{(j,i) : sim[j][i][j] > thr}   (s5:89)
 * The comparison is the reference's fp32 `S > thres`; the counts are exact.  FAR/FRR, with the reference's own
 * denominators (s5:81,88), and the argmin over thresholds (s5:93-98) are a few dozen scalar operations on the host. */
int ge2e_eer_counts(const float* sim, int B, int N, int M, const float* thresholds, int T, int* counts, void* stream);

/* Batch sampler (s1_dataset_loader.py:65-77, EmbeddingModelTTDataset.__getitem__ for the N speakers of a batch): the
 * per-speaker spectrogram arrays (U_j, T, F), float64 as the reference stores them on disk (sv_<speaker>.npy) or float32,
 * stay resident in ONE device buffer `store`; spk_offsets [N] (device, int64, in ELEMENTS) locate the arrays of the batch's
 * speakers, utter_idx [N][M] and clip_start [N] (device, int32) are the draws s1:65 and s1:71 make on the host.
 *   out [N][M][L][F] float32 = (float) array_n[utter_idx[n][m]][clip_start[n] .. + L][:]      (s1:68, 74; the cast of s2:28)
 * The caller guarantees 0 <= utter_idx < U_n and 0 <= clip_start <= T - L (the reference's draws do); N * M <= 65535. */
int ge2e_sample_batch(const void* store, int store_is_f64, const long long* spk_offsets, const int* utter_idx,
                      const int* clip_start, int N, int M, int T, int L, int F, float* out, void* stream);

/* ---- diagnostics (tests only): device building blocks on caller data ------------------- */
/* A,Bm [64][256], G [64][64] (|x| <= 1) -> X [64][64] = A.Bm^T, GE [64][256] = G.A,
 * GC [64][256] = G^T.Bm through the split-fp16 MFMA tile contractions. */
int ge2e_selftest_split_gemm(const float* A, const float* Bm, const float* G,
                             float* X, float* GE, float* GC, void* stream);
/* x [64] -> out [384]: wave_sum, wave_max, row16_sum, quad_sum, wave_argmax idx, quad_argmax idx. */
int ge2e_selftest_wave_ops(const float* x, float* out, void* stream);

/* One wave of the team kernel's per-speaker chain on caller data: CH [64][256], R [16][256] (unit rows)
 * -> XT [64][16] = CH.R^T (16x16x32 tiles), GE [16][256] = XT^T.CH with XT taken straight from the
 * accumulators, GT = GE stored through the in-quad register transpose. */
int ge2e_selftest_rows16(const float* CH, const float* R, float* XT, float* GE, float* GT, void* stream);
/* Team formation (8 workgroups of one XCD) + the L2 hand-off protocol under load: `grid` workgroups
 * (all resident), `rounds` publish/consume rounds of `payload_f4` float4 per member.
 * out [16] (device): [0] complete teams, [1] mismatching float4 read back, [2..9] workgroups per XCD,
 * [10] abort word.  ws: ge2e_selftest_team_bytes(payload_f4) bytes, 256-byte aligned. */
size_t ge2e_selftest_team_bytes(int payload_f4);
int ge2e_selftest_team(void* ws, size_t ws_bytes, int grid, int rounds, int payload_f4, unsigned* out,
                       void* stream);

/* ge2e_loss_fwd_bwd with impl = GE2E_IMPL_TEAM and the team kernel's abort word raised before the launch: exercises the
 * in-launch redo (the same workgroups, one per batch) deterministically.  Same arguments and results. */
int ge2e_selftest_team_fallback(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                                float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dE,
                                float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* The same with the abort word raised by ONE workgroup (block index grid / 2) when it reaches the end of the launch: the
 * workgroups that finished before leave, the later ones stay, and the redo must agree on its size (TeamCtl::go).  dE may be
 * NULL (the forward-only team kernel).  Same arguments and results. */
int ge2e_selftest_team_abort_midgrid(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                                     float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dE,
                                     float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ge2e_loss_fwd_bwd with impl = GE2E_IMPL_TEAM on at most max_workgroups workgroups (>= 64, rounded down to a multiple
 * of 64 = one team per XCD): the same results from fewer teams, i.e. many more batches through each team's hand-off
 * counters than a full-size launch reaches. */
int ge2e_selftest_team_grid(const float* E, int B, int N, int M, int D, const float* w, const float* b,
                            float eps_cos, float eps, int variant, float* loss, float* per_emb_loss, float* dE,
                            float* dw, float* db, void* workspace, size_t workspace_bytes, void* stream,
                            int max_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* GE2E_HIP_H */
