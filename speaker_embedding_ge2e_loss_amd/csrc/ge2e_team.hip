// GE2E_IMPL_TEAM: eight workgroups on eight CUs of one XCD share a batch; E is read from HBM ONCE and every
// phase works on flat 16-row blocks of the member's rows, so nothing is padded from M rows to an MFMA block.
//
// Member m of a team owns the speaker slots 8 m .. 8 m + spm - 1 (spm = ceil(N / 8)) and their R = spm M rows.
// LDS holds only the member's OWN data: its rows as fp16 hi / lo unit-row images ET (87 KB at N=64, M=10, D=256),
// the similarity block X (two K-halves, fp32) and the dL/dS images G.  The 64 unit centroids never enter LDS: each
// wave keeps the MFMA fragments it needs in registers, loaded straight from the team's L2-resident exchange area
// (row-major for X = ET . CH^T, and as 16-byte k-groups per column for gE = G . CH).
//
// One iteration starts batch n ("cur") and finishes batch n - 1 ("prev"); phases in program order:
//   A1   wave s = speaker slot s: sum of its M rows (prefetched registers) -> unit centroid -> published in both forms;
//        drain, barrier, signal: prev's partial gradients (hand-off 2) and cur's centroids (hand-off 1)
//   A2   the wave's rows -> |e|, e-hat -> ET images (the hand-off travels meanwhile)
//   W    wait for both hand-offs
//   B    centroid fragments from L2 -> registers; wave (t, kh): X[all rows][slots 16 t ..] over the K half kh -> LDS
//        (60 MFMAs a wave, SIMD-balanced)
//   S    softmax / contrast on X, 16 lanes per row: leave-one-out statistics, loss, dL/dS -> G images, row coefficients
//   F    FINISH of prev: the seven other members' partial gradients of my speaker (+ my own slice, kept in LDS) -> KJ_j;
//        of cur only a scalar is left (KJP_j = kjb c-hat_j: the leave-one-out speaker row's e-hat part rides in gC, see S)
//   dE   dE(prev) = held + KJ_j -> HBM in whole 128-byte lines (the only write of dE)
//   GC   partial gC^T[d][k] = sum_r ET[r][d] G[r][k] (32 x 32 x 16 tiles) -> published for the next iteration's F
//   --   the next batch's rows requested (the only read of E)
//   GE   gE[d][r] = sum_k CH[k][d] G[r][k] with the centroid fragments from registers; ra gE + c1 e-hat is held in
//        registers (40 VGPRs) until the partial gradients of the other members arrive
// The own-speaker column of a row carries the coefficient of s_j in the G image (so GE adds that term for free); what
// that entry adds to the member's own partial gC IS, pushed through the centroid norm, the leave-one-out speaker row
// sum_i c3_i e-hat_i up to a multiple of c-hat_j (see S): nothing has to be taken out or added as a vector.
//
// Exchange per batch and team: 128 KB of centroids (two layouts, double-buffered by batch parity) + 448 KB of partial
// gradients in ONE buffer guarded by a read-done counter (c3); measured fabric traffic 1.4x the algorithmic bytes.
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_team.hpp"
#include "ge2e_team_kernel.hpp"
#include "ge2e_team_dev.hpp"
#include "ge2e_fused.hpp"
#include <atomic>

namespace ge2e {




// ---------------------------------------------------------------------------------------------
static int team_cu_count() { return device_cu_count(); }

// LDS bytes of one X half-block / of the G images for `rt` image rows (host layout and the compile-time-shape kernels)
constexpr unsigned team_xb_bytes(int rt, int D) {
    unsigned xb = (unsigned)rt * XP * 4;               // X half-blocks; the first also holds the KJ rows [8][D] of phase F
    if (xb < 16u * D * 4) xb = 16u * D * 4;            // + the member's own slice of its partial gC [8][D]; the second stages
    if (xb < 2u * 8 * (D + STGPAD) * 2) xb = 2u * 8 * (D + STGPAD) * 2;   // the centroid for its k-group form: [hi, lo][8][D + 32] halfs
    return xb;
}
constexpr unsigned team_g_bytes(int rt, int D) {     // the G images; between two batches the region holds the KJ rows [8][D]
    return 2u * rt * GP * 2 > 8u * D * 4 ? 2u * rt * GP * 2 : 8u * D * 4;
}

TeamKWs team_layout(int N, int M, int Dc) {
    const int D = (Dc + 63) / 64 * 64;     // the kernels' column count: the caller's D padded to a multiple of 64
    TeamKWs L{};
    L.spm = (N + TEAM - 1) / TEAM;
    L.rt = (L.spm * M + 15) / 16 * 16;
    L.mul_m = (65536 + M - 1) / M;
    L.head_bytes = (unsigned)team_head_bytes();
    static_assert(sizeof(TeamKFlags) == 3 * 128 && team_head_bytes() >= sizeof(TeamCtl) + 64 * sizeof(TeamKFlags), "head layout");
    const size_t et = (size_t)2 * L.rt * D * 2;
    const size_t xb = team_xb_bytes(L.rt, D);
    const size_t g = team_g_bytes(L.rt, D);
    L.xb_bytes = (unsigned)xb;
    L.g_bytes = (unsigned)g;
    L.lds_bytes = et + 2 * xb + g + (size_t)(L.rt * 8 + NC * 4 + 32 + 16) * sizeof(float);
    return L;
}

bool team_supports(int N, int M, int D) {
    if (!(N >= 1 && N <= NC && M >= 2 && M <= 16 && D >= 4 && D <= 256 && (D % 4) == 0)) return false;
    const TeamKWs L = team_layout(N, M, D);
    if (L.rt > RTMAX || L.lds_bytes > 160 * 1024) return false;
    if (!fused_split_supports(N, M, D)) return false;            // the gated fall-back launch
    return team_cu_count() >= MAX_XCD * TEAM;                    // partitioned device: no XCD-wide teams to form
}

// workgroups: one per CU, but no more than eight XCDs' worth of teams for the batches there are
int team_grid(int B) {
    const int cus = team_cu_count() / (MAX_XCD * TEAM) * (MAX_XCD * TEAM);
    const long want = (long)((B + MAX_XCD - 1) / MAX_XCD) * (MAX_XCD * TEAM);
    return (int)(want < cus ? want : cus);
}
// The fall-back runs at the one-workgroup-per-batch kernel's own grid (round 2 capped it at 32 workgroups to keep the
// workspace small: an eighth of the chip when a large launch had to fall back).
int team_fallback_grid(int B) {
    const int cus = team_cu_count();
    return B < cus ? B : cus;
}
// one slice per workgroup of the team grid (a redo is shared by team_fallback_grid(B) of them; a workgroup whose end-of-grid
// wait ran out redoes the call alone in ITS slice)
static size_t team_fb_bytes(int B, int N, int M, int D) {
    const int g = team_grid(B) > team_fallback_grid(B) ? team_grid(B) : team_fallback_grid(B);
    return (size_t)g * fused_split_layout(N, M, D).stride * sizeof(float);
}
size_t team_workspace_bytes(int B, int N, int M, int D) {
    const TeamKWs L = team_layout(N, M, D);
    return align_up(L.head_bytes + (size_t)(team_grid(B) / TEAM) * team_exchange((D + 63) / 64 * 64).stride, 256) + team_fb_bytes(B, N, M, D);
}

// ---------------------------------------------------------------------------------------------
// D = 64 * NCH; MR >= M rows of a speaker are prefetched into registers; RBT > 0: the member's images have exactly
// RBT 16-row blocks (compile-time trip counts for the metric shape), RBT == 0: L.rt / 16 at run time.
// FWD: the forward-only instantiation of the metric shape (similarity + loss, dE == NULL known at compile time: no held /
// fragment / partial-gradient registers, no gradient phases in the loop at all).
template <int NCH, int MR, int RBT, bool CONTRAST, bool FWD = false>
__global__ __launch_bounds__(512, 2) void ge2e_team_kernel(Problem p, TeamKWs L, FusedWs F) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int D = 64 * NCH;
    constexpr int P = D;                  // image pitch: no padding, chunks swizzled by row (et_off)
    constexpr unsigned ROWB = D * 4;      // bytes per row of the exchange / LDS layouts (D = 64 NCH columns)
    // The caller's D may be any multiple of 4 up to 64 NCH (AUTO pads it here rather than falling through to the VALU kernel):
    // rows of E and dE are DG floats apart, columns DG .. D - 1 are read as zeros and never stored.
    const int DG = RBT ? D : p.D;
    const unsigned ROWBG = (unsigned)DG * 4u;
    constexpr int NT = 4 * NCH;           // 16-column tiles of a row
    constexpr int NTI = (NT + 7) / 8;     // ... per wave in GE
    constexpr TeamKX XO = team_exchange(D);
    constexpr int RBC = RBT ? RBT : RBMAX;
    const int RB = RBT ? RBT : L.rt / 16;
    const int RT = 16 * RB;
    const int RBr = RB;
    constexpr bool CT_X = RBT != 0, CT_S = RBT != 0, CT_DE = RBT != 0, CT_GC = RBT != 0;
    _Float16* const ETh = reinterpret_cast<_Float16*>(smem_f);
    _Float16* const ETl = ETh + RT * P;
    float* const XB0 = reinterpret_cast<float*>(ETl + RT * P);
    // compile-time shape: every LDS offset is an immediate (fewer scalars to keep -- the kernel spills SGPRs)
    const unsigned xb_bytes = RBT ? team_xb_bytes(16 * RBT, D) : L.xb_bytes;
    const unsigned g_bytes = RBT ? team_g_bytes(16 * RBT, D) : L.g_bytes;
    float* const XB1 = XB0 + xb_bytes / 4;
    _Float16* const Gh = reinterpret_cast<_Float16*>(XB1 + xb_bytes / 4);
    _Float16* const Gl = Gh + RT * GP;
    float* const RS = reinterpret_cast<float*>(reinterpret_cast<char*>(Gh) + g_bytes);   // [RT][8]
    float* const CST = RS + RT * 8;                                // [64][4]  1/|c|, kappa, |s|, |s|^2 of every slot
    float* const RED = CST + NC * 4;                               // [32]
    int* const SH = reinterpret_cast<int*>(RED + 32);              // [16]
    float* const KJ = reinterpret_cast<float*>(Gh);                // F1 -> dE: KJ_j rows [8][D]; the G images are dead from GE(prev) to S(cur)
    float* const OWNP = XB0 + 8 * D;                               // GC -> next A1: own slots of this member's partial gC [8][D]
    _Float16* const STG = reinterpret_cast<_Float16*>(XB1);        // A: centroid stage [hi, lo][8 slots][D + STGPAD]
    constexpr int SP = D + STGPAD;

    // the metric-shape instantiation (RBT != 0) is launched only with M == MR: row loops lose their `i < M` branches
    // and N == 64 (every member full: 8 speakers, 80 rows), so that the has-a-speaker predicates fold away too
    constexpr bool MEX = RBT != 0;
    const int N = MEX ? 64 : p.N, M = MEX ? MR : p.M, NM = N * M;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    TeamCtl* const ctl = reinterpret_cast<TeamCtl*>(p.ws);
    TeamKFlags* const flags = reinterpret_cast<TeamKFlags*>(ctl + 1);
    const TeamId id = team_form(ctl, SH, p.launch_seq);
    if (id.team == -2) {    // a control block that cannot be trusted: no counters at all -- static redo, workgroup 0 leaves a clean block
        team_redo<NCH>(p, L, F, smem_f, (int)gridDim.x, (int)blockIdx.x);
        __syncthreads();
        if (blockIdx.x == 0 && tid < 64) team_head_rewrite(reinterpret_cast<unsigned*>(ctl), (int)(L.head_bytes / 16), 1u, p.launch_seq);
        return;
    }
    if (id.nct == 0 && blockIdx.x == 0 && tid == 0)   // no eight workgroups share an XCD: the call is redone at the end of this launch
        __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the team part as a block of its own: workgroups without a team skip it and meet the others in team_finish below)
    bool gave_up = id.nct == 0;     // this workgroup knows the call will be redone (no team anywhere; below: its own hand-off ran out)
    [&]() __attribute__((always_inline)) {
    if (id.team < 0) return;
    TeamKFlags* const fl = flags + id.team;
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(
        reinterpret_cast<const char*>(p.ws) + L.head_bytes + (size_t)id.team * XO.stride, XO.stride);

    const int spm = MEX ? 8 : L.spm;
    const int j0 = id.member * spm;                    // first speaker of this member
    const int my_spm = MEX ? 8 : max(0, min(spm, N - j0));
    const int R_my = my_spm * M;
    const bool has_spk = wid < my_spm;
    const int j = j0 + wid;                            // this wave's speaker (if has_spk)
    const int kslot = 8 * id.member + wid;             // ... and the slot every wave is responsible for in A
    const int rbase = wid * M;                         // first row of that speaker in the images

    // w and b are read once and kept as SCALAR bit patterns; every phase derives the constants it needs (w log2 e, 1 / M,
    // ...) from an opaque copy of them (GE2E_T2_CONSTS).  Derived once at kernel scope they are uniform values in VGPRs
    // for the whole kernel -- a dozen registers the allocator spills around the contractions.
    const int w_bits = __builtin_amdgcn_readfirstlane(__float_as_int(p.w ? *p.w : p.w_imm));
    const int b_bits = __builtin_amdgcn_readfirstlane(__float_as_int(p.b ? *p.b : p.b_imm));
    const float eps = p.eps, eps_cos = p.eps_cos;
#define GE2E_T2_CONSTS()                                                                       \
    int wb_ = w_bits, bb_ = b_bits, mc_ = M, le_ = __float_as_int(p.log_eps);                  \
    asm volatile("" : "+s"(wb_), "+s"(bb_), "+s"(mc_), "+s"(le_));                             \
    const float w = __int_as_float(wb_), bias = __int_as_float(bb_);                           \
    const float eps_cos2 = eps_cos * eps_cos;                                                  \
    const float fM = (float)mc_, inv_m = rcp_nr(fM), inv_m1 = rcp_nr((float)(mc_ - 1));       \
    (void)w; (void)bias; (void)eps_cos2; (void)fM; (void)inv_m; (void)inv_m1; (void)le_
    const bool want_grad = !FWD && p.dE != nullptr;
    const int tX = wid & 3, khX = wid >> 2;            // X: slot tile and K half of this wave
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

    // rows of the ET images that never receive an embedding stay zero (they are contracted over in GC)
    for (int i = tid; i < RT * P / 8; i += 512) {
        reinterpret_cast<float4*>(ETh)[i] = zero4();
        reinterpret_cast<float4*>(ETl)[i] = zero4();
    }
    for (int i = tid; i < RT * 2; i += 512) reinterpret_cast<float4*>(RS)[i] = zero4();   // rows without an embedding
    // ... and so do their rows of the G images (S writes only the rows of speakers that exist)
    for (int i = tid; i < (int)(g_bytes / 16); i += 512) reinterpret_cast<float4*>(Gh)[i] = zero4();
    __syncthreads();

    float4 rowv[MR];            // this wave's rows of the batch about to start
    float4 held[NTI][RBC];      // ra gE + c1 e-hat of rows 16 rb + l15, columns 16 dt + 4 q ..   (GE -> next iteration)
    float kjb = 0.f;            // speaker row KJP_j = kjb c-hat_j: its coefficient                (F2 -> next F1)
    h8 ga[NTI][2][2] = {};      // GE's centroid fragments: columns 16 dt + l15, slots 32 s2 + 8 q .. (k-group form; GC -> next GE)
    float4 cj_cur = zero4(), cj_prev = zero4();   // c-hat_j, this lane's 4 columns
    float rn_cur = 0.f, kap_cur = 0.f, rn_prev = 0.f, kap_prev = 0.f;
#pragma unroll
    for (int i = 0; i < NTI; ++i)
#pragma unroll
        for (int rb = 0; rb < RBC; ++rb) held[i][rb] = zero4();

#define GE2E_T2_LOAD_ROWS(BI)                                                                            \
    do {                                                                                                 \
        int lq_ = lane;                                                                                  \
        asm volatile("" : "+v"(lq_));                                                                    \
        const unsigned vrow_ = 4 * lq_ < DG ? (unsigned)lq_ * 16u : OOB;                                 \
        const bool on_ = has_spk && (BI) < p.B;                                                          \
        const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(p.E + (size_t)(on_ ? (BI) : 0) * NM * DG, (unsigned)NM * ROWBG); \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                   \
            rowv[i] = bload4<GE2E_T2_E_AUX>(rs_, (on_ && i < M) ? vrow_ : OOB, (unsigned)(j * M + min(i, M - 1)) * ROWBG); \
    } while (0)
    // lane-derived indices are re-derived inside each phase from an opaque copy of the lane id (kept out of the
    // loop-invariant set: hoisted they spill, and a scratch reload queues behind every VMEM operation in flight)
#define GE2E_T2_LANE()                                                  \
    int lv_ = lane;                                                     \
    asm volatile("" : "+v"(lv_));                                       \
    const int l15 = lv_ & 15, q = lv_ >> 4, d4 = 4 * lv_;               \
    const bool dact = D == 256 || d4 < D;   /* 64 lanes x 4 columns: every lane has columns at D = 256 */ \
    (void)l15; (void)q; (void)d4; (void)dact

    GE2E_PROF_DECL(20)
#ifdef GE2E_PROF_A1     // finer view of A1 (tools/profile_phases.py prints slots 14..18)
#define GE2E_PROF_SUB(i) GE2E_PROF(i)
#else
#define GE2E_PROF_SUB(i)
#endif
    GE2E_T2_LOAD_ROWS(id.team);
    const TeamId id_outer = id;
    int wslot = 0;
    bool failed = false;
    const int wid_outer = wid, member_outer = id.member, tid_outer = tid, m_outer = M;
    for (int seq = 0;; ++seq) {
        // wave- and member-derived scalars are re-derived in every iteration from opaque copies: as loop invariants hipcc
        // precomputes ~130 of them in front of the loop, spills them to VGPR lanes and reads them back one v_readlane at
        // a time (184 per iteration and wave); an s_mul / s_add where the value is needed is cheaper
        int wid_o = wid_outer, mem_o = member_outer, tid_o = tid_outer, m_o = m_outer;
        asm volatile("" : "+s"(wid_o), "+s"(mem_o), "+v"(tid_o), "+s"(m_o));
        const int wid = wid_o, tid = tid_o, lane = tid & 63, M = MEX ? MR : m_o, NM = N * M;
        TeamId id = id_outer;
        id.member = mem_o;
        const int j0 = id.member * spm;
        const int my_spm = MEX ? 8 : max(0, min(spm, N - j0));
        const int R_my = my_spm * M;
        const bool has_spk = wid < my_spm;
        const int j = j0 + wid;
        const int kslot = 8 * id.member + wid;
        const int rbase = wid * M;
        const int tX = wid & 3, khX = wid >> 2;
        const int bi = id.team + seq * id.nct;          // batch started in this iteration
        const bool have_cur = bi < p.B, have_prev = seq > 0;
        if (!have_cur && !have_prev) break;
        const int buf = seq & 1, pbuf = buf ^ 1;
        const __amdgpu_buffer_rsrc_t rsGp = make_rsrc(want_grad && have_prev ? p.dE + (size_t)(bi - id.nct) * NM * DG : nullptr,
                                                       want_grad && have_prev ? (unsigned)NM * ROWBG : 0u);
        cj_prev = cj_cur; rn_prev = rn_cur; kap_prev = kap_cur;
        // ===== A1(cur): speaker sum -> unit centroid -> team (row-major, and staged for the k-group form) ==========
        if (have_cur) {
            GE2E_T2_LANE();
            GE2E_T2_CONSTS();
            float4 s = zero4();
#pragma unroll
            for (int i = 0; i < MR; ++i)
                if (i < M) { s.x += rowv[i].x; s.y += rowv[i].y; s.z += rowv[i].z; s.w += rowv[i].w; }
            GE2E_PROF_SUB(14);
            const float4 c = scale4(s, inv_m);
            float sqs[2] = {dot4(c, c), dot4(s, s)};
            wave_sum_to_sgpr<2>(sqs);
            const float sq = sqs[0], ss = sqs[1];
            float rn, kap, nc;
            unit_stats_bf(sq, eps_cos, eps_cos2, rn, kap, nc);
            if (!has_spk) { rn = 0.f; kap = 0.f; nc = 0.f; }     // slots without a speaker publish zero rows
            cj_cur = has_spk ? scale4(c, rn) : zero4();
            rn_cur = rn; kap_cur = kap;
            h4 hi, lo;
            split4_scaled(cj_cur, kSplitScale, hi, lo);
            {   // fragment-major form for X: one 1-KB block per (slot tile, hi / lo, 32-column K-step) holding the 64 lanes'
                // 16-byte MFMA fragments in lane order (lane = 16 q + slot-in-tile), so that a consumer's load instruction
                // reads 1 KB contiguously.  (As plain row-major rows every load instruction fetched 16 x 64 bytes, the
                // shape the texture path likes least; eight of those per wave and batch.)  This lane holds columns d4 ..
                // d4 + 3 of slot kslot: half a fragment.
                const unsigned blk = (unsigned)(kslot >> 4) * (4u * NCH) + (unsigned)(lv_ >> 3);      // hi block of its K-step
                const unsigned vh = dact ? blk * 1024u + (unsigned)(((lv_ >> 1) & 3) * 16 + (kslot & 15)) * 16u + (unsigned)(lv_ & 1) * 8u : OOB;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rsX, vh + XO.chr[buf], 0, GE2E_T2_XC_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rsX, vh + XO.chr[buf] + 2048u * NCH, 0, GE2E_T2_XC_AUX);
            }
            if (dact && want_grad) {   // one 8-byte write per image: a row of the stage per wave (the k-group form is gE's operand)
                *reinterpret_cast<h4*>(STG + wid * SP + d4) = hi;
                *reinterpret_cast<h4*>(STG + (8 + wid) * SP + d4) = lo;
            }
            // 1/max(|c|,eps), kappa, |s_j| (s_j = c-hat_j * that), |s_j|^2
            bstore4(rsX, lane == 0 ? XO.cst[buf] + (unsigned)kslot * 16u : OOB,
                                    make_float4(rn, kap, has_spk ? fM * nc : 0.f, has_spk ? ss : 0.f));
        }
        GE2E_PROF_SUB(15);
        if (want_grad) __syncthreads();      // (forward only: no stage, no k-group form -- one barrier and 64 KB of exchange fewer per batch)
        GE2E_PROF_SUB(16);
        if (want_grad && have_cur && tid < 2 * D) {   // the k-group form: 16 bytes (this member's 8 slots) per (hi / lo, d), gathered
            // by the transposing LDS read (4 slots x 16 columns per 16 lanes, twice); whole waves only (2 D % 64 == 0)
            const int hl = tid >= D, d = tid - hl * D;
            const int l16 = tid & 15;
            const _Float16* p0 = STG + (8 * hl + (l16 >> 2)) * SP + (d - l16) + 4 * (l16 & 3);
            const h4 a = tr_read4(p0), b2 = tr_read4(p0 + 4 * SP);
            const h8 v = __builtin_shufflevector(a, b2, 0, 1, 2, 3, 4, 5, 6, 7);
            bstore4<GE2E_T2_XC_AUX>(rsX, XO.cht[buf] + (unsigned)id.member * (2u * D * 16u) + (unsigned)tid * 16u, __builtin_bit_cast(float4, v));
        }
        // ---- one drain + barrier publishes prev's partial gradients (hand-off 2) and cur's centroid (hand-off 1)
        GE2E_PROF_SUB(17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GE2E_PROF_SUB(18);
        __syncthreads();
        if (tid == 0) {
            if (have_prev) add_agent(&fl->c2, 1u);
            if (have_cur) add_agent(&fl->c1, 1u);
        }
        GE2E_PROF(0);

        if (want_grad && have_prev) {
            // ===== GE(prev): gE^T[d][r] = sum_k CH[k][d] G[r][k]; ra gE + c1 e-hat stays in registers until dE ==========
            // The G fragments of the next row block and this block's epilogue operands are requested before the MFMAs.
            {
                GE2E_T2_LANE();
                int rbg = RB;
                asm volatile("" : "+s"(rbg));
                h8 gb[2][2][2];             // [set][s2][hi, lo]: G rows 16 rb + l15, slots 32 s2 + 8 q ..
#define T2_GE_LOAD(RB_)                                                                                      \
    do {                                                                                                     \
        const int r_ = 16 * (RB_) + l15;                                                                     \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                   \
            gb[(RB_) & 1][s2][0] = frag_row(Gh + g_off(r_, 32 * s2 + 8 * q));                                 \
            gb[(RB_) & 1][s2][1] = frag_row(Gl + g_off(r_, 32 * s2 + 8 * q));                                 \
        }                                                                                                    \
    } while (0)
                T2_GE_LOAD(0);
#pragma unroll
                for (int rb = 0; rb < RBC; ++rb) {
                    if (rb < rbg) {
                        const int r = 16 * rb + l15;
                        if (rb > 0) T2_GE_LOAD(rb);
                        const float2 rc = *reinterpret_cast<const float2*>(RS + r * 8 + 4);   // ra, c1
                        h4 eh[NTI], el[NTI];
#pragma unroll
                        for (int i = 0; i < NTI; ++i) {
                            const int eo = et_off<D>(r, min(16 * T2_DT(i), D - 16) + 4 * q);
                            eh[i] = *reinterpret_cast<const h4*>(ETh + eo);
                            el[i] = *reinterpret_cast<const h4*>(ETl + eo);
                        }
                        // lane (r = l15, q) ends with gE[r][16 dt + 4 q + i].  
#pragma unroll
                        for (int i = 0; i < NTI; ++i) {
                            f32x4 acc = acc_zero4();
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2)
                                mfma16x3(acc, ga[i][s2][0], ga[i][s2][1], gb[rb & 1][s2][0], gb[rb & 1][s2][1]);
                            // ra acc + c1 (e-hat hi + e-hat lo): the fp16 image values enter a mixed-precision FMA directly
                            // (v_fma_mix_f32: the same fma(float(h), c1, t) as a convert and an fmaf, in one instruction)
                            const uint2 ehb = __builtin_bit_cast(uint2, eh[i]), elb = __builtin_bit_cast(uint2, el[i]);
                            held[i][rb] = make_float4(fma_mix_lo(ehb.x, rc.y, fma_mix_lo(elb.x, rc.y, acc[0] * rc.x)),
                                                      fma_mix_hi(ehb.x, rc.y, fma_mix_hi(elb.x, rc.y, acc[1] * rc.x)),
                                                      fma_mix_lo(ehb.y, rc.y, fma_mix_lo(elb.y, rc.y, acc[2] * rc.x)),
                                                      fma_mix_hi(ehb.y, rc.y, fma_mix_hi(elb.y, rc.y, acc[3] * rc.x)));
                        }
                        if (NTI == 2) T2_PAIR_LINES(held[0][rb], held[1][rb]);
                        __builtin_amdgcn_sched_barrier(0);   // one row block at a time (registers)
                    }
                }
#undef T2_GE_LOAD
            }
        }
        GE2E_PROF(10);

        // ===== W: both hand-offs (signalled by every member at the same point) ======================================
        {
            int* const wsh = SH + 4 + (wslot & 3);
            ++wslot;
            if (tid == 0) {
                bool ok = true;
                if (have_cur) ok = spin_until(&fl->c1, (unsigned)(TEAM * (seq + 1)), ctl);
                if (ok && have_prev) ok = spin_until(&fl->c2, (unsigned)(TEAM * seq), ctl);
                *wsh = ok ? 1 : 0;
            }
            __syncthreads();                 // also: every wave's ET rows and row scalars are written
            if (*wsh == 0) { failed = true; break; }
        }
        GE2E_PROF(2);

        // ===== B: centroid fragments, previous batch's partial gradients and scalars -> registers ===================
        h8 xa[NCH][2];          // X: slots 16 tX + l15, K-steps khX NCH + s, 8 q ..      (row-major form)
        float4 part[TEAM - 1];  // the seven other members' partials of my speaker (prev), in rotated member order
        float4 cstv, scv;
        {   // every load unconditional (out-of-bounds offsets read zeros): a load under an `if` is a phi at the join, and
            // hipcc resolves it with s_waitcnt vmcnt(0) -- a full L2 round trip in front of A2
            GE2E_T2_LANE();
            // the partial gradients FIRST: F1 needs them right behind A2, the centroid fragments only at the X contraction,
            // and the memory queue returns in issue order
            {   // my speaker's partial gradients of prev (first iteration, forward only: out of bounds, zeros); used in F1
                const unsigned vrow = (dact && have_prev && has_spk && want_grad) ? (unsigned)d4 * 4u : OOB;
#pragma unroll
                for (int mm = 1; mm < TEAM; ++mm)
                    part[mm - 1] = bload4<AUX_L2>(rsX, vrow + XO.gc + (unsigned)(((id.member + mm) & (TEAM - 1)) * NC + kslot) * ROWB, 0);
            }
            // block (slot tile tX, hi / lo, K-step khX NCH + s) of the fragment-major centroid form, this lane's 16 bytes
            const unsigned oa = have_cur ? XO.chr[buf] + ((unsigned)tX * (4u * NCH) + (unsigned)khX * NCH) * 1024u + (unsigned)lv_ * 16u : OOB;
#pragma unroll
            for (int s = 0; s < NCH; ++s) {
                xa[s][0] = bload_h8<AUX_L2>(rsX, oa + 1024u * s, 0);
                xa[s][1] = bload_h8<AUX_L2>(rsX, oa + 1024u * s + 2048u * NCH, 0);
            }
            cstv = bload4<AUX_L2>(rsX, have_cur && tid < NC ? XO.cst[buf] + (unsigned)tid * 16u : OOB, 0);
            scv = bload4<AUX_L2>(rsX, have_prev && id.member == 0 && tid < TEAM ? XO.sc[pbuf] + (unsigned)tid * 16u : OOB, 0);
        }
        GE2E_PROF(9);


        // ===== A2(cur): own rows -> |e|, e-hat -> ET images (the hand-off travels meanwhile) ========================
        if (have_cur && has_spk) {
            GE2E_T2_LANE();
            GE2E_T2_CONSTS();
            // The rows' squared norms are reduced together into scalars and dropped into LANE i of one register; the norm
            // bookkeeping then runs ONCE (lane i = row i) instead of once per row on all 64 lanes, the row scalars leave
            // in one LDS write, and each row's scale factor comes back as a scalar.
            float eev[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) eev[i] = i < M ? dot4(rowv[i], rowv[i]) : 0.f;
            // row i's |e|^2 ends in lane scatter_lane(i) of ONE register (ge2e_common.hpp: wave_sums_scatter)
            const float ee_l = wave_sums_scatter<MR>(eev, lv_);
            float rne_l, ke_l, ne_l;
            unit_stats_bf(ee_l, eps_cos, eps_cos2, rne_l, ke_l, ne_l);
            {   // the lane that holds row i's scalars writes them: i = 4 (lane & 15) + [0, 2, 1, 3][lane >> 4]
                const int rho = lv_ >> 4, irow = 4 * (lv_ & 15) + (((rho & 1) << 1) | (rho >> 1));
                if ((lv_ & 15) < (MR + 3) / 4 && irow < M)
                    *reinterpret_cast<float4*>(RS + (rbase + irow) * 8) = make_float4(rne_l, ke_l, ee_l, ne_l);
            }
            const float rs_l = rne_l * kSplitScale;
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                if (i < M) {
                    const float sc = lane_get(rs_l, scatter_lane(i));
                    if (dact) {     // scale and split fused (v_fma_mixlo / mixhi_f16: 8 instructions per float4 instead of 16)
                        h4 hi4, lo4;
                        split4_scaled_u(rowv[i], sc, hi4, lo4);
                        const int eo = et_off<D>(rbase + i, d4);
                        *reinterpret_cast<h4*>(ETh + eo) = hi4;
                        *reinterpret_cast<h4*>(ETl + eo) = lo4;
                    }
                }
                if (i & 1) __builtin_amdgcn_sched_barrier(0);   // two rows at a time (registers)
            }
        }
        // The next batch's rows are requested as soon as this batch's have become images: a whole iteration ahead of their
        // first use (A1 at the top of the next one), and BEHIND the requests above in the memory queue, which retires in
        // order: nothing that is waited for before the next A1 queues behind this HBM round trip.
        GE2E_T2_LOAD_ROWS(bi + id.nct);
        GE2E_PROF(1);

        // ===== F1: KJ_j of prev from the partial gradients (requested before A2, in registers by now) -> LDS ========
        // Straight-line on purpose: in the first iteration (no prev) the loads went out of bounds and read zeros; only the
        // counter add and the KJ store are conditional.  (With the loads and the sums under separate `if (have_prev)`s
        // hipcc split the loop body on that flag, put the loads a region away from their uses and parked them in scratch.)
        if (want_grad && has_spk) {
            GE2E_T2_LANE();
            GE2E_T2_CONSTS();
            // fixed order (own slice first -- it never travels through L2 --, then members member+1 .. member+7 mod 8)
            float4 gsum = *reinterpret_cast<const float4*>(OWNP + wid * D + min(d4, D - 4));
            if (!have_prev) gsum = zero4();
#pragma unroll
            for (int m = 0; m < TEAM - 1; ++m) { gsum.x += part[m].x; gsum.y += part[m].y; gsum.z += part[m].z; gsum.w += part[m].w; }
            // every partial of this speaker has been read (the sums above waited for them): the single gC buffer may be
            // rewritten
            asm volatile("" :: "v"(gsum.x), "v"(gsum.y), "v"(gsum.z), "v"(gsum.w));
            if (lane == 0 && have_prev) add_agent(&fl->c3, 1u);
            gsum = scale4(gsum, w * kSplitInv2);
            float cf[1] = {dot4(gsum, cj_prev)};
            wave_sum_to_sgpr<1>(cf);
            const float coefc = cf[0];
            const float sc = rn_prev * inv_m, fk = kjb - kap_prev * coefc * sc;   // KJ_j = sc gC_j + (kjb - kap (gC_j . c-hat_j) sc) c-hat_j
            if (dact && have_prev)
                *reinterpret_cast<float4*>(KJ + wid * D + d4) =
                    make_float4(fmaf(gsum.x, sc, fk * cj_prev.x), fmaf(gsum.y, sc, fk * cj_prev.y),
                                fmaf(gsum.z, sc, fk * cj_prev.z), fmaf(gsum.w, sc, fk * cj_prev.w));
        }
        __syncthreads();                 // KJ rows; every wave's ET rows and row scalars of cur are written
        GE2E_PROF(5);

        // ===== dE_r of prev = held part + KJ_{speaker of r}: two or three speakers per 16-row block ================
        // All KJ reads first, then the sums and stores: written as one loop hipcc recycles ONE temporary and serialises
        // ten LDS round trips per wave.  The sums go to fresh registers: formed in place (held += KJ) under a two-trip
        // loop, every trip carried the forty held registers through phi copies -- 60 v_mov_b64 per wave and batch.
        // Metric shape (M and the row count are compile-time): the speaker of row 16 rb + 8 i + x, x = l15 & 7, is
        // (16 rb + 8 i) / M + (x >= M - (16 rb + 8 i) % M): a compile-time KJ row plus one of three lane offsets (0 or one
        // KJ row), so a read is a base register and an immediate.  Tried and dropped: the stores under GC's MFMAs -- the
        // wave blocks at store issue and GC went from 6 k to 10 k cycles (-7 % overall).
#define T2_DE_STORES()                                                                                                     \
    do {                                                                                                                   \
            GE2E_T2_LANE();                                                                                                \
            const int x8_ = l15 & 7;                                                                                       \
            const float* const kjc_ = KJ + min(NTI == 2 ? 32 * wid + 16 * (l15 >> 3) + 4 * q : 4 * q, D - 4);             \
            const int dcol_ = NTI == 2 ? 32 * wid + 16 * (l15 >> 3) + 4 * q : 4 * q;   /* + cc: this lane's first column */ \
            const unsigned de0_ = (unsigned)((j0 * M + (NTI == 2 ? x8_ : l15)) * DG + dcol_) * 4u;                         \
_Pragma("unroll")                                                                                                          \
            for (int i = 0; i < NTI; ++i) {                                                                                \
                float4 kjv[RBC];                                                                                           \
_Pragma("unroll")                                                                                                          \
                for (int rb = 0; rb < RBC; ++rb)                                                                           \
                    if (CT_DE || rb < RBr) {                                                                               \
                        if (MEX) {                                                                                         \
                            const int b0 = 16 * rb + 8 * i;                                                                \
                            kjv[rb] = *reinterpret_cast<const float4*>(kjc_ + (b0 / MR) * D + (x8_ >= MR - b0 % MR ? D : 0)); \
                        } else {                                                                                           \
                            const int r = NTI == 2 ? 16 * rb + 8 * i + x8_ : 16 * rb + l15;                                \
                            const int c = NTI == 2 ? 0 : 16 * T2_DT(i);                                                    \
                            kjv[rb] = *reinterpret_cast<const float4*>(kjc_ + min((r * L.mul_m) >> 16, 7) * D + min(c, D - 16)); \
                        }                                                                                                  \
                    }                                                                                                      \
_Pragma("unroll")                                                                                                          \
                for (int rb = 0; rb < RBC; ++rb)                                                                           \
                    if (CT_DE || rb < RBr) {                                                                               \
                        const int rc = NTI == 2 ? 16 * rb + 8 * i : 16 * rb;      /* row and column part that is not in de0_ */ \
                        const int cc = NTI == 2 ? 0 : 16 * T2_DT(i);                                                       \
                        const bool ok = MEX || (rc + (NTI == 2 ? x8_ : l15) < R_my && T2_DT(i) < NT && dcol_ + cc < DG);   \
                        const float4 o = make_float4(held[i][rb].x + kjv[rb].x, held[i][rb].y + kjv[rb].y,                 \
                                                     held[i][rb].z + kjv[rb].z, held[i][rb].w + kjv[rb].w);                \
                        bstore4<GE2E_T2_DE_AUX>(rsGp, ok ? de0_ + (unsigned)(rc * DG + cc) * 4u : OOB, o);                 \
                    }                                                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                         \
            }                                                                                                              \
    } while (0)
        // ===== X(cur): X[r][slot] over this wave's K half -> LDS (fragments of the next row block under the MFMAs) ==
#define T2_X_LOAD(T_)                                                                                     \
    do {                                                                                                  \
        const _Float16* const pp_ = ETh + xs[(T_) % NCH] + 16 * ((T_) / NCH) * P;                         \
        fb[(T_) & 1][0] = frag_row(pp_);                                                                  \
        fb[(T_) & 1][1] = frag_row(pp_ + RT * P);                                                         \
    } while (0)
#define T2_X_STORE(RB_)                                                                       \
    *reinterpret_cast<float4*>(XBk + (16 * (RB_) + l15) * XP + 16 * tX + 4 * q) =             \
        make_float4(acc[(RB_) & 1][0], acc[(RB_) & 1][1], acc[(RB_) & 1][2], acc[(RB_) & 1][3])
#define T2_X_CONTRACT()                                                                                                    \
    do {                                                                                                                   \
            GE2E_T2_LANE();                                                                                                \
            float* const XBk = khX ? XB1 : XB0;                                                                            \
            const int fx = (4 * (l15 & 3) + ((4 - (l15 >> 2)) & 3)) & ((D % 128 == 0) ? 15 : 7);   /* et_off's f of rows 16 rb + l15 */ \
            h8 fb[2][2];            /* [K-step parity][hi, lo]: the next K-step's row fragments are requested under this one's MFMAs */ \
            int xs[NCH];            /* one lane offset per K-step (the swizzle has period 16 in the row); row block + lo image: immediates */ \
            _Pragma("unroll") for (int s2 = 0; s2 < NCH; ++s2) xs[s2] = l15 * P + (((4 * (khX * NCH + s2) + q) ^ fx) << 3); \
            f32x4 acc[2] = {acc_zero4(), acc_zero4()};                                                                     \
            T2_X_LOAD(0);                                                                                                  \
_Pragma("unroll")                                                                                                          \
            for (int rb = 0; rb < RBC; ++rb) {                                                                             \
                if (CT_X || rb < RBr) {                                                                                    \
                    acc[rb & 1] = acc_zero4();                                                                             \
_Pragma("unroll")                                                                                                          \
                    for (int s = 0; s < NCH; ++s) {                                                                        \
                        const int t = rb * NCH + s;                                                                        \
                        if (t + 1 < RBC * NCH && (CT_X || t + 1 < RBr * NCH)) T2_X_LOAD(t + 1);                            \
                        mfma16x3(acc[rb & 1], xa[s][0], xa[s][1], fb[t & 1][0], fb[t & 1][1]);                             \
                        __builtin_amdgcn_sched_barrier(0);   /* fragments at most one K-step ahead (registers) */          \
                    }                                                                                                      \
                    /* lane (r = l15, q) holds X[16 tX + 4 q + i][16 rb + l15]; the previous block's sums are final now */ \
                    if (rb > 0) { T2_X_STORE(rb - 1); }                                                                    \
                }                                                                                                          \
            }                                                                                                              \
            if (CT_X) { T2_X_STORE(RBT - 1); }                                                                             \
            else {                                                                                                         \
_Pragma("unroll")                                                                                                          \
                for (int rb = 0; rb < RBC; ++rb)                                                                           \
                    if (rb == RBr - 1) { T2_X_STORE(rb); }                                                                 \
            }                                                                                                              \
            if (tid < NC) *reinterpret_cast<float4*>(CST + tid * 4) = cstv;                                                \
    } while (0)
        // The stores of a workgroup leave at ~14 B/clk, 6 k cycles for all eight waves, and a wave blocks while its
        // stores wait to issue.  dE(prev) and X(cur) are independent (the stores read KJ and the held registers, X reads
        // the images and writes the X block): the two waves of a SIMD take them in OPPOSITE order -- waves 0-3 store while
        // waves 4-7 contract, then the other way round -- so a wave blocked at store issue shares its SIMD with one that
        // feeds the matrix pipe.  Both orders are written out (a two-trip loop around one copy made phis of everything
        // that is live across it).
        if (wid < 4) {
            if (want_grad && have_prev) T2_DE_STORES();
            if (have_cur) T2_X_CONTRACT();
        } else {
            if (have_cur) T2_X_CONTRACT();
            if (want_grad && have_prev) T2_DE_STORES();
        }
#undef T2_X_CONTRACT
#undef T2_X_LOAD
#undef T2_X_STORE
#undef T2_DE_STORES
        GE2E_PROF(6);
        // ---- previous batch: scalars out ----------------------------------------------------------------------------
        if (have_prev) {
            if (id.member == 0 && wid == 0) {
                const float l = oct_sum(scv.x), a = oct_sum(scv.y), c = oct_sum(scv.z);
                if (lane == 0) {
                    if (p.loss) p.loss[bi - id.nct] = l;
                    if (p.dw) p.dw[bi - id.nct] = a;
                    if (p.db) p.db[bi - id.nct] = c;
                }
            }
        }
        if (!have_cur) break;

        GE2E_PROF(11);
        __syncthreads();
        GE2E_PROF(3);

        // ===== S(cur): leave-one-out statistics, S, loss, dL/dS -> G images, row coefficients =======================
        // Wave s = speaker slot s (as in A): FOUR lanes per row, 16 similarities per lane, so the per-row algebra is
        // replicated 4x instead of 16x, the row reductions are two quad DPP steps and one pass covers the speaker.
        // Lane (rr = lane >> 2, qq = lane & 3) holds the slots (sb + j) & 63, j = 0..15, sb = (own slot & ~3) + 16 qq:
        // aligned groups of four, and the own-speaker column is always one of values 0..3 of the lane qq == 0.
        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        float c4v = 0.f;                // row coefficient c4' of row i in lane 4 i (S -> F2, same wave: no LDS trip)
        if (have_cur && want_grad && R_my < RT) {   // image rows without a speaker: the KJ rows of F1 have been lying there
            for (int i = tid; i < (RT - R_my) * GP / 8; i += 512) {
                reinterpret_cast<float4*>(Gh + R_my * GP)[i] = zero4();
                reinterpret_cast<float4*>(Gl + R_my * GP)[i] = zero4();
            }
        }
        if (have_cur && has_spk) {
            GE2E_T2_LANE();
            GE2E_T2_CONSTS();
            const float w2 = w * LOG2E, b2 = (w * eps + bias) * LOG2E, leps2 = __int_as_float(le_) * LOG2E;   // softmax in base 2
            const int rr = lv_ >> 2, qq = lv_ & 3;
            const bool rv = rr < M;
            const int r = rbase + min(rr, M - 1);
            const int ko = kslot;                                   // own-speaker slot (wave-uniform)
            const int sb = (ko & ~3) + 16 * qq;
            const int jo = ko & 3;
            const bool own_lane = qq == 0;
            const bool all_valid = N == NC;
            float x[16];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int s4 = (sb + 4 * jj) & (NC - 1);
                const float4 a = *reinterpret_cast<const float4*>(XB0 + r * XP + s4);
                const float4 b = *reinterpret_cast<const float4*>(XB1 + r * XP + s4);
                x[4 * jj + 0] = a.x + b.x; x[4 * jj + 1] = a.y + b.y; x[4 * jj + 2] = a.z + b.z; x[4 * jj + 3] = a.w + b.w;
            }
            const float xo = (XB0[r * XP + ko] + XB1[r * XP + ko]) * kSplitInv2;   // c-hat_j . e-hat_r
            const float4 rs0 = *reinterpret_cast<const float4*>(RS + r * 8);    // rne ke ee |e|
            const float4 cs = *reinterpret_cast<const float4*>(CST + ko * 4);   // rn kap |s| |s|^2
            const float rne = rs0.x, ke = rs0.y, ee = rs0.z, ne = rs0.w;
            const float es = xo * cs.z * ne;                 // e . s_j
            const float eu = (es - ee) * inv_m1;
            const float uu = fmaxf((cs.w - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
            float rnu, ku, nu;
            unit_stats_bf(uu, eps_cos, eps_cos2, rnu, ku, nu);
            const float cosd = eu * rne * rnu;               // cos(e, leave-one-out centroid)
            const float sjj2 = fmaf(w2, cosd, b2);
            const float w2s = w2 * kSplitInv2;               // S2 = w2s x + b2 on the raw accumulator sums
            // validity of this lane's slots (members with fewer than spm speakers): slot = 8 m' + loc valid iff loc < nval(m')
            auto vld = [&](int j) {
                const int s = (sb + j) & (NC - 1);
                return all_valid || (s & 7) < max(0, min(spm, N - (s >> 3) * spm));
            };
            bool ownj[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ownj[j] = own_lane && jo == j;
            float per, ad0, coefsum, db_row;
            float gv[16];
            float gsc = 0.f, own_un = 0.f;     // per-lane scale of the G image entries; factor that puts o on the unscaled side
            if (!CONTRAST) {
                // the row maximum over the OTHER speakers' columns (the own column enters with its leave-one-out value)
                float xm0 = ownj[0] ? x[1] : x[0], xm1 = ownj[1] ? x[0] : x[1], xm2 = ownj[2] ? x[3] : x[2], xm3 = ownj[3] ? x[2] : x[3];
                float xm;
                if (w2 >= 0.f) {
                    xm = fmaxf(fmaxf(xm0, xm1), fmaxf(xm2, xm3));
#pragma unroll
                    for (int j = 4; j < 16; j += 2) xm = fmaxf(fmaxf(x[j], x[j + 1]), xm);
                } else {
                    xm = fminf(fminf(xm0, xm1), fminf(xm2, xm3));
#pragma unroll
                    for (int j = 4; j < 16; j += 2) xm = fminf(fminf(x[j], x[j + 1]), xm);
                }
                float mx = quad_max(fmaf(w2s, xm, b2));
                mx = fmaxf(fmaxf(mx, sjj2), leps2);
                const float t = b2 - mx;
#pragma unroll
                for (int j = 0; j < 16; ++j) gv[j] = __builtin_amdgcn_exp2f(fmaf(w2s, x[j], t));
                if (!all_valid) {   // members with fewer than spm speakers: their empty slots do not count
#pragma unroll
                    for (int j = 0; j < 16; ++j) gv[j] = vld(j) ? gv[j] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] = ownj[j] ? 0.f : gv[j];
                float zl = 0.f, al = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    zl += gv[j];
                    al = fmaf(gv[j], x[j], al);
                }
                const float zp = quad_sum(zl);                // sum over the other speakers, shifted
                const float ap = quad_sum(al);
                const float zoff = zp + __builtin_amdgcn_exp2f(leps2 - mx);
                const float z = zoff + __builtin_amdgcn_exp2f(sjj2 - mx);
                per = LN2 * ((mx - sjj2) + __builtin_amdgcn_logf(z));
                const float rz = rcp_nr(z);
                ad0 = rv ? -zoff * rz : 0.f;                  // dL/dS on the own-speaker column: -(1 - p_jj) = -z_off / z
                const float rzv = rv ? rz : 0.f;
                coefsum = fmaf(ap * kSplitInv2, rzv, ad0 * cosd);     // sum_k dL/dS_k c0_k
                db_row = fmaf(zp, rzv, ad0);
                gsc = rzv * kSplitScale;                       // the G image holds gv * gsc (fused into the split below)
                own_un = z;                                    // own column: o * z * (rz 2^8) = o 2^8 (1 +- 2^-23)
            } else {
                float best = -INFINITY, bx = 0.f; int besti = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int s = (sb + j) & (NC - 1);
                    const bool ok = vld(j) && !(j < 4 && ownj[j & 3]);
                    const float sve = ok ? fmaf(w2s, x[j], b2) : -INFINITY;
                    if (ok && (sve > best || (sve == best && s < besti))) { best = sve; besti = s; bx = x[j]; }
                }
                const int loci = besti;
                quad_argmax(best, besti);
                bx = quad_sum(loci == besti && besti != 0x7fffffff ? bx : 0.f);
                const float pos = rcp_nr(1.0f + __builtin_amdgcn_exp2f(-sjj2));
                const float neg = (N > 1) ? rcp_nr(1.0f + __builtin_amdgcn_exp2f(-best)) : 0.0f;
                per = 1.0f - pos + neg;
                ad0 = rv ? -pos * (1.0f - pos) : 0.f;
                const float gn = rv ? neg * (1.0f - neg) : 0.f;
                coefsum = fmaf(gn, bx * kSplitInv2, ad0 * cosd);
                db_row = gn + ad0;
#pragma unroll
                for (int j = 0; j < 16; ++j) gv[j] = (((sb + j) & (NC - 1)) == besti && vld(j)) ? gn : 0.f;
                gsc = kSplitScale;
                own_un = 1.0f;
            }
            if (rv && own_lane) {
                loss_acc = per;
                dw_acc = fmaf(eps, db_row, coefsum);
                db_acc = db_row;
                if (p.per) p.per[(size_t)bi * NM + j0 * M + rbase + rr] = per;
            }
            if (want_grad) {
                const float coef = w * coefsum;                  // (dL/d e-hat) . e-hat, own-speaker term included
                // dE_r = ra acc + c1 e-hat + c2 s_j + KJ_j   (ge2e_fused_f32.hip header for the algebra); everything
                // that multiplies the own-column gradient is linear in w, so the G image can carry the coefficient
                // of s_j (o, in units of ra) without dividing by w
                const float ad = w * ad0;
                const float rho = rnu * inv_m1;
                const float t1 = ku * cosd * rho;                        // kappa_u cos rho
                const float c2_0 = rho * ad0 * (rne + t1);               // c2 / w
                const float c1 = -ke * coef * rne - ad * rho - w * c2_0 * ne;
                const float alpha = ad * rnu * (1.0f + t1 * ne);
                const float beta = -ad * rnu * t1;
                const float o = c2_0 * cs.z * ne;
                // o also lands in this member's partial gC of slot ko: W = w sum_i o_i e-hat_i.  KJ_j is linear in gC, and
                // pushed through the centroid norm W is EXACTLY the leave-one-out speaker row sum_i c3_i e-hat_i
                // (c3 = alpha / (M - 1); (rn_j / M) w o = c3 because rn_j |s_j| / M = 1 and rne |e| = 1) minus
                // kap_j (sum_i c3_i xo_i) c-hat_j.  So the e-hat part of KJP_j is already in gC and only a multiple of
                // c-hat_j is left: c4' = c4 + kap_j c3 xo.  (Rounds 2-3 formed c3' = c3 - (rn_j / M) w o -- rounding
                // noise around zero -- and spent 190 instructions per wave and batch on sum_i c3'_i e-hat_i.)
                const float c4p = inv_m1 * (beta * cs.z + cs.y * alpha * xo);       // c4' (of c-hat_j)
                c4v = rv && own_lane ? c4p : 0.f;
                if (rv && own_lane)
                    *reinterpret_cast<float2*>(RS + r * 8 + 4) =
                        make_float2(rne * (w * kSplitInv2),                         // ra: of the gE accumulator (2^16)
                                    c1 * kSplitInv);                                // c1: of the e-hat image value (2^8)
                const float og = o * own_un;
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[j] = ownj[j] ? og : gv[j];
                if (rv) {
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        h4 gh4, gl4;
                        split4_scaled(make_float4(gv[4 * jj], gv[4 * jj + 1], gv[4 * jj + 2], gv[4 * jj + 3]), gsc, gh4, gl4);
                        const int go = g_off(r, (sb + 4 * jj) & (NC - 1));
                        *reinterpret_cast<h4*>(Gh + go) = gh4;
                        *reinterpret_cast<h4*>(Gl + go) = gl4;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- member scalars: fixed-order reduction over the 8 waves ------------------------------------------------
        loss_acc = wave_sum(loss_acc);
        dw_acc = wave_sum(dw_acc);
        db_acc = wave_sum(db_acc);
        if (lane == 0) { RED[wid] = loss_acc; RED[8 + wid] = dw_acc; RED[16 + wid] = db_acc; }
        __syncthreads();
        GE2E_PROF(4);

        // GE's centroid fragments (k-group form, published in this iteration's A1) are requested HERE, in front of the two
        // phases that issue no memory instruction (F2 is a scalar reduction now, GC's contraction reads LDS): the address
        // path is idle now and saturated behind GC's contraction, where these eight 1-KB loads per wave used to stand in
        // front of the read-done poll and the 32 partial-gradient stores.  (32 registers live through F2 and GC.)
#define T2_GA_LOAD()                                                                                                      \
    do {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NTI; ++i) {                                                                 \
            const int dt = T2_DT(i);                                                                                      \
            const bool on = dt < NT;                                                                                      \
            _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                            \
                const unsigned o = XO.cht[buf] + (unsigned)(4 * s2 + q) * (2u * D * 16u) + (unsigned)(16 * dt + l15) * 16u; \
                ga[i][s2][0] = bload_h8<AUX_L2>(rsX, on ? o : OOB, 0);                                                    \
                ga[i][s2][1] = bload_h8<AUX_L2>(rsX, on ? o + (unsigned)D * 16u : OOB, 0);                                \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)
        if (want_grad) { GE2E_T2_LANE(); T2_GA_LOAD(); }
        // ===== F2: member scalars out; the coefficient of KJP_j = kjb c-hat_j of cur (kept in a register until the next F1) =
        if (have_cur && tid == 0) {
            float l = 0.f, a = 0.f, c = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { l += RED[i]; a += RED[8 + i]; c += RED[16 + i]; }
            bstore4(rsX, XO.sc[buf] + (unsigned)id.member * 16u, make_float4(l, a, c, 0.f));
        }
        if (want_grad && has_spk) {   // speaker row KJP_j = (sum_i c4'_i) c-hat_j: one scalar, kept for the next F1
            float bs[1] = {c4v};
            wave_sum_to_sgpr<1>(bs);
            kjb = bs[0];
        }
        GE2E_PROF(12);
        if (want_grad) {
            // ===== GC: partial gC[k][d] = sum_r G[r][k] ET[r][d]; wave: slots 32 kh.., columns 64 sl.. ===============
            {
                GE2E_T2_LANE();
                const int kh = wid >> 2, sl = wid & 3;
                if (64 * sl < D) {
                    f32x16 gc[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < 16; ++i) gc[b][i] = 0.f;
                    h8 gf[2][2], ef[2][2][2];     // [set][hi, lo], [set][b][hi, lo]
                    // Both swizzles are periodic in 16 rows (g_off: 8), so a K-step's fragment addresses are the K-step-0
                    // addresses plus a compile-time row offset: six lane offsets per phase and immediates after that.  (Written
                    // with the K-step inside g_off / et_off, hipcc re-derived every swizzle per step: 190 integer instructions
                    // per wave and batch in this phase alone.)
                    const int r0_ = 8 * (lv_ >> 5) + ((lv_ & 15) >> 2), c0_ = 16 * ((lv_ >> 4) & 1) + 4 * (lv_ & 3);
                    const int gA = g_off(r0_, 32 * kh + c0_), gB = g_off(r0_ + 4, 32 * kh + c0_);
                    const int eA[2] = {et_off<D>(r0_, 64 * sl + c0_), et_off<D>(r0_, 64 * sl + 32 + c0_)};
                    const int eB[2] = {et_off<D>(r0_ + 4, 64 * sl + c0_), et_off<D>(r0_ + 4, 64 * sl + 32 + c0_)};
#define T2_TR8(IMG_, A_, B_, ROWS_) \
    __builtin_shufflevector(tr_read4((IMG_) + (A_) + (ROWS_)), tr_read4((IMG_) + (B_) + (ROWS_)), 0, 1, 2, 3, 4, 5, 6, 7)
#define T2_GC_LOAD(S_)                                                                   \
    do {                                                                                 \
        gf[(S_) & 1][0] = T2_TR8(Gh, gA, gB, 16 * (S_) * GP);                            \
        gf[(S_) & 1][1] = T2_TR8(Gl, gA, gB, 16 * (S_) * GP);                            \
        _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                  \
            ef[(S_) & 1][b][0] = T2_TR8(ETh, eA[b], eB[b], 16 * (S_) * P);               \
            ef[(S_) & 1][b][1] = T2_TR8(ETl, eA[b], eB[b], 16 * (S_) * P);               \
        }                                                                                \
    } while (0)
                    T2_GC_LOAD(0);
#pragma unroll
                    for (int s = 0; s < RBC; ++s) {
                        if (CT_GC || s < RBr) {
                            if (s + 1 < RBC && (CT_GC || s + 1 < RBr)) T2_GC_LOAD(s + 1);
#pragma unroll
                            for (int b = 0; b < 2; ++b) mfma32x3(gc[b], gf[s & 1][0], gf[s & 1][1], ef[s & 1][b][0], ef[s & 1][b][1]);
                            __builtin_amdgcn_sched_barrier(0);   // fragments at most one K-step ahead (registers)
                        }
                    }
#undef T2_GC_LOAD
#undef T2_TR8
                    // the single partial-gradient buffer: the previous batch's partials must have been read by everybody
                    bool ok = true;
                    if (seq > 0) {
                        int okv = 1;
                        if (lv_ == 0) okv = spin_until(&fl->c3, (unsigned)(N * seq), ctl) ? 1 : 0;
                        ok = __builtin_amdgcn_readfirstlane(okv) != 0;
                    }
                    if (ok) {
                        // lane (d = l31, h): register i = slot 32 kh + (i & 3) + 8 (i >> 2) + 4 h, column 64 sl + 32 b + l31.
                        // One dword per lane: every store instruction writes two whole 128-byte row segments (a 16-byte-
                        // per-lane store of this accumulator would scatter 32-byte pieces over 32 rows, and every piece
                        // leaves the L2 as a write request of its own)
                        const int l31 = lv_ & 31, h = lv_ >> 5;
                        const unsigned ob = XO.gc + (unsigned)((id.member * NC + 32 * kh + 4 * h) * D + 64 * sl + l31) * 4u;
                        // One lane offset for all 32 stores, the (register, column half) part in the scalar offset; the four
                        // registers of the member's own slots -- 32 kh + 8 (member & 3) + (i & 3) + 4 h, they stay in LDS
                        // (below) -- are skipped by a uniform branch.  (A per-store `own ? OOB : ob + c` was a v_add and a
                        // v_cndmask per store: 64 VALU instructions per wave and batch behind the contraction.)
                        const int own_g = kh == (id.member >> 2) ? (id.member & 3) : -1;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            if (g != own_g) {
#pragma unroll
                                for (int b = 0; b < 2; ++b)
#pragma unroll
                                    for (int t = 0; t < 4; ++t)
                                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gc[b][4 * g + t]), rsX, ob,
                                                                              (unsigned)((t + 8 * g) * D + 32 * b) * 4u, GE2E_T2_GC_AUX);
                            }
                        }
                        if (kh == (id.member >> 2)) {
                            float* const o = OWNP + (4 * h) * D + 64 * sl + l31;
#define T2_OWN(G_)                                                                                  \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                   \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) o[t * D + 32 * b] = gc[b][4 * (G_) + t];
                            switch (id.member & 3) {
                                case 0: T2_OWN(0) break;
                                case 1: T2_OWN(1) break;
                                case 2: T2_OWN(2) break;
                                default: T2_OWN(3) break;
                            }
#undef T2_OWN
                        }
                    }
                } else {
                }
            }
#undef T2_GA_LOAD
            GE2E_PROF(7);
        }
    }
    if (failed && tid == 0) __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gave_up = gave_up || failed;
    GE2E_PROF_FLUSH(20)
    }();
    // ---- end of the launch (team_finish, ge2e_team.hpp): one load and one atomic in the common case, the last workgroup hands
    //      the control block back clean; with the abort word up the workgroups that are still there redo the call with the
    //      one-workgroup-per-batch body
    {
        if (p.test_abort == 2 && blockIdx.x == (gridDim.x >> 1)) {   // diagnostics: the word rises in the middle of the grid's finish
            if (tid == 0) __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
        }
        int* const fsh = reinterpret_cast<int*>(smem_f) + 4;     // (LDS is free now: every wave of this workgroup is here)
        const TeamRedo rd = team_finish(ctl, fsh, (int)(L.head_bytes / 16), gave_up);
        if (rd.n != 0) {
            team_redo<NCH>(p, L, F, smem_f, rd.n, rd.rank);
            team_redo_done(ctl, fsh, (int)(L.head_bytes / 16), rd.n);
        }
    }
#undef GE2E_PROF_SUB
#undef GE2E_T2_LOAD_ROWS
#undef GE2E_T2_LANE
#undef GE2E_T2_CONSTS
}

// ---------------------------------------------------------------------------------------------
// a clean control block (zeros + magic) at the head of a workspace: n16 16-byte pieces; raise_abort: the abort word comes up 1
__global__ __launch_bounds__(256) void team_zero_head(uint4* head, int n16, int raise_abort) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i == (int)(offsetof(TeamCtl, magic) / 16)) v.x = TEAM_MAGIC;
        if (raise_abort && i == (int)(offsetof(TeamCtl, abort_) / 16)) v.x = 1u;
        head[i] = v;
    }
}
hipError_t launch_team_head_init(void* head, size_t bytes, bool raise_abort, hipStream_t stream) {
    const int n16 = (int)(bytes / 16);
    hipLaunchKernelGGL(team_zero_head, dim3((n16 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<uint4*>(head), n16,
                       raise_abort ? 1 : 0);
    return hipGetLastError();
}

template <int NCH, int MR, int RBT, bool CONTRAST, bool FWD = false>
static hipError_t launch_nch(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(ge2e_team_kernel<NCH, MR, RBT, CONTRAST, FWD>);
    static KernelLaunchState state;     // one per instantiation; per-device entries inside
    int nb = 0;
    hipError_t err = prepare_kernel(state, fn, 512, (unsigned)L.lds_bytes, &nb);
    if (err != hipSuccess) return err;
    // ONE launch per call: the control block is self-cleaning (ge2e_team.hpp) -- the previous call's last workgroup left it
    // clean, ge2e_workspace_init wrote a first clean one, and anything else makes this launch redo itself without teams
    // and leave a clean block.  (p.test_abort: diagnostics, a block with the abort word raised is written in front.)
    if (p.test_abort == 1) {
        err = launch_team_head_init(p.ws, L.head_bytes, true, stream);
        if (err != hipSuccess) return err;
    }
    // Every workgroup must be resident (they wait for each other): grid <= resident capacity is what a cooperative
    // launch checks; the same check is made here and the kernel goes out as an ordinary launch.  Should the teams
    // not form, or a bounded spin run out, the kernel raises the control block's abort word and its own workgroups
    // redo every batch with the one-workgroup-per-batch body once the whole grid has met (team_finish).
    int grid = team_grid(p.B);
    if (p.grid_cap > 0 && p.grid_cap < grid)      // diagnostics: fewer teams, more batches through each (whole XCD rounds)
        grid = p.grid_cap / (MAX_XCD * TEAM) * (MAX_XCD * TEAM) > 0 ? p.grid_cap / (MAX_XCD * TEAM) * (MAX_XCD * TEAM) : MAX_XCD * TEAM;
    if (nb < 1 || grid > nb * team_cu_count()) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((ge2e_team_kernel<NCH, MR, RBT, CONTRAST, FWD>), dim3(grid), dim3(512), L.lds_bytes, stream, p, L, F);
    return hipGetLastError();
}
template <int NCH, int MR>
static hipError_t launch_variant(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream) {
    if (NCH == 4 && MR == 10 && L.rt == 80 && p.M == 10 && p.N == 64 && p.D == 256) {   // the metric shape: compile-time N, M, D, trip counts
        if (p.dE == nullptr)    // ... and its forward-only form (evaluation: s4:61-110, s5:42-44)
            return p.variant == 1 ? launch_nch<4, 10, 5, true, true>(p, L, F, stream) : launch_nch<4, 10, 5, false, true>(p, L, F, stream);
        return p.variant == 1 ? launch_nch<4, 10, 5, true>(p, L, F, stream) : launch_nch<4, 10, 5, false>(p, L, F, stream);
    }
    return p.variant == 1 ? launch_nch<NCH, MR, 0, true>(p, L, F, stream) : launch_nch<NCH, MR, 0, false>(p, L, F, stream);
}

hipError_t launch_team(const Problem& p_in, hipStream_t stream) {
    Problem p = p_in;
    {   // TeamCtl::gen: a number per launch, never 0 (a HIP graph replays the number it captured: a replay that finds an
        // untrusted block redoes itself without counters every time until another launch or ge2e_workspace_init cleans it)
        static std::atomic<unsigned> seq{0};
        unsigned s = seq.fetch_add(1u, std::memory_order_relaxed) + 1u;
        p.launch_seq = s ? s : seq.fetch_add(1u, std::memory_order_relaxed) + 1u;
    }
    TeamKWs L = team_layout(p.N, p.M, p.D);
    const FusedWs F = fused_split_layout(p.N, p.M, p.D);
    // the redo body's slices behind the teams' exchange areas; its LDS if that is larger than the team kernel's
    L.fb_off = align_up(L.head_bytes + (size_t)(team_grid(p.B) / TEAM) * team_exchange((p.D + 63) / 64 * 64).stride, 256);
    L.fb_wgs = team_fallback_grid(p.B);
    if (fused_split_lds_bytes(p.D) > L.lds_bytes) L.lds_bytes = fused_split_lds_bytes(p.D);
    hipError_t err;
    if (p.dE == nullptr) {          // similarity + loss only: the pipelined forward kernel (ge2e_team_fwd.hip)
        err = launch_team_fwd(p, L, F, stream);
    } else if (p.M <= 10) {
        switch ((p.D + 63) / 64) {
            case 1: err = launch_variant<1, 10>(p, L, F, stream); break;
            case 2: err = launch_variant<2, 10>(p, L, F, stream); break;
            case 3: err = launch_variant<3, 10>(p, L, F, stream); break;
            default: err = launch_variant<4, 10>(p, L, F, stream); break;
        }
    } else {
        switch ((p.D + 63) / 64) {
            case 1: err = launch_variant<1, 16>(p, L, F, stream); break;
            case 2: err = launch_variant<2, 16>(p, L, F, stream); break;
            case 3: err = launch_variant<3, 16>(p, L, F, stream); break;
            default: err = launch_variant<4, 16>(p, L, F, stream); break;
        }
    }
    return err;
}

}  // namespace ge2e
