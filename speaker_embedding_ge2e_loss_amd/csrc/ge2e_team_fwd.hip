// GE2E_IMPL_TEAM, FORWARD ONLY (dE == NULL): similarity + loss -- the launch of the reference's evaluation paths (s4:61-110 test
// loss, s5:42-44) and the workload north_star's roofline sentence names.  A kernel of its own, not the training pipeline
// with the gradient phases compiled out (rounds 3-4): that one left a forward batch as ONE serial chain per workgroup --
// speaker sums -> publish -> hand-off -> images -> contraction -> softmax, five workgroup barriers, nothing else resident on
// the CU to fill the waits (14.5 k cycles per batch for ~6 k cycles of issue).
//
// Same team as ge2e_team.hip (eight workgroups on eight CUs of one XCD share a batch, E is read from HBM once, member m
// owns speaker slots 8 m .. and their rows), same exchange area and control block.  What differs is the schedule: TWO
// batches are in flight per workgroup and every long latency has most of an iteration to travel.
//
//   iteration n (cur = batch n of this team, prev = n - 1):
//     A1(cur)   wave s = speaker slot s: sum of its M rows (registers, requested two iterations ago) -> unit centroid ->
//               published as MFMA fragments (fragment-major, ge2e_team.hip); prev's fragments requested underneath
//     X(prev)   X[slot][row] of prev on 16x16x32 split-fp16 MFMA: centroid fragments of prev in registers (requested at
//               the end of the previous iteration), the member's e-hat images in LDS; wave (slot tile, K half) -> XB0 / XB1
//     --        drain, BARRIER 1, one lane signals c1: cur's centroids are published (and prev's scalars, below)
//     A2(cur)   wave s: |e| of its rows (one reduce-scatter), e-hat -> split-fp16 images (overwrites prev's: X(prev) is done)
//     rows      the rows of batch n + 2 requested into the registers A2 has just emptied (two register sets alternate)
//     S(prev)   wave = speaker, 4 lanes per row, 16 similarities per lane from XB0 + XB1: leave-one-out cosine on the own
//               column, softmax / contrast, per-row loss -> loss, (dw, db) partials
//     W         one lane polls c1 for cur (signalled a phase and a half ago); BARRIER 2
//     scalars   member scalars of prev -> exchange; member 0 sums the eight members' scalars of batch n - 2 (made visible by
//               this iteration's hand-off) -> loss / dw / db.  (cur's centroid fragments are requested from L2 at the top of
//               the next iteration, one load at a time between the instruction groups of its A1.)
//   Two workgroup barriers per batch.  HBM latency has the whole iteration (rows), the hand-off has A2 + S, the L2 round
//   trip of the fragments has A1.  After the last batch two extra signals carry the last two batches' scalars to member 0.
//
// LDS per member: e-hat hi / lo images 80 KB + X halves 43.5 KB + row scalars (double-buffered) 5 KB: one workgroup per CU.
// Exchange per batch and team: 64 KB of centroid fragments (double-buffered by parity) + 128 bytes of scalars.
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_team.hpp"
#include "ge2e_team_kernel.hpp"
#include "ge2e_team_dev.hpp"

namespace ge2e {

// LDS of the forward-only kernel for `rt` image rows
size_t team_fwd_lds_bytes(int rt, int D) {
    return (size_t)2 * rt * D * 2 + (size_t)2 * rt * XP * 4 + (size_t)(2 * rt * 4 + 32 + 16) * sizeof(float);
}

template <int NCH, int MR, int RBT, bool CONTRAST>
__global__ __launch_bounds__(512, 2) void ge2e_team_fwd_kernel(Problem p, TeamKWs L, FusedWs F) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int D = 64 * NCH;
    constexpr int P = D;                  // image pitch: no padding, chunks swizzled by row (et_off)
    const int DG = RBT ? D : p.D;         // the caller's D (a multiple of 4, <= D): rows of E are DG floats apart, the rest zeros
    const unsigned ROWBG = (unsigned)DG * 4u;
    constexpr TeamKX XO = team_exchange(D);
    constexpr unsigned SC4 = XO.cst[0];   // member scalars [4 batches in flight][8 members][4 floats]: the region the training
                                          // kernel uses for slot scalars (2 KB), which the forward pass keeps in registers
    constexpr int RBC = RBT ? RBT : RBMAX;
    const int RB = RBT ? RBT : L.rt / 16;
    const int RT = 16 * RB;
    const int RBr = RB;
    constexpr bool CT_X = RBT != 0;
    _Float16* const ETh = reinterpret_cast<_Float16*>(smem_f);
    _Float16* const ETl = ETh + RT * P;
    float* const XB0 = reinterpret_cast<float*>(ETl + RT * P);
    float* const XB1 = XB0 + RT * XP;
    float* const RS = XB1 + RT * XP;                               // [2][RT][4]  1/|e|c, kappa, |e|^2, |e|c  (by batch parity)
    float* const RED = RS + 2 * RT * 4;                            // [32]
    int* const SH = reinterpret_cast<int*>(RED + 32);              // [16]

    constexpr bool MEX = RBT != 0;        // the metric shape: N = 64, M = MR, every member full
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
    const int w_bits = __builtin_amdgcn_readfirstlane(__float_as_int(p.w ? *p.w : p.w_imm));
    const int b_bits = __builtin_amdgcn_readfirstlane(__float_as_int(p.b ? *p.b : p.b_imm));
    const float eps = p.eps, eps_cos = p.eps_cos;
#define GE2E_TF_CONSTS()                                                                       \
    int wb_ = w_bits, bb_ = b_bits, mc_ = M, le_ = __float_as_int(p.log_eps);                  \
    asm volatile("" : "+s"(wb_), "+s"(bb_), "+s"(mc_), "+s"(le_));                             \
    const float w = __int_as_float(wb_), bias = __int_as_float(bb_);                           \
    const float eps_cos2 = eps_cos * eps_cos;                                                  \
    const float fM = (float)mc_, inv_m = rcp_nr(fM), inv_m1 = rcp_nr((float)(mc_ - 1));       \
    (void)w; (void)bias; (void)eps_cos2; (void)fM; (void)inv_m; (void)inv_m1; (void)le_
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

    // rows of the images that never receive an embedding stay zero (finite similarities for the padded rows)
    for (int i = tid; i < RT * P / 8; i += 512) {
        reinterpret_cast<float4*>(ETh)[i] = zero4();
        reinterpret_cast<float4*>(ETl)[i] = zero4();
    }
    for (int i = tid; i < 2 * RT; i += 512) reinterpret_cast<float4*>(RS)[i] = zero4();
    __syncthreads();

    float4 ra[MR], rb2[MR];     // this wave's rows: RA = the batch about to start, RB = the one after it (in flight)
    float sn_cur = 0.f, ss_cur = 0.f, sn_prev = 0.f, ss_prev = 0.f;   // |s_j| (clamped), |s_j|^2 of this wave's speaker
    // every load unconditional: a disabled one gets an out-of-bounds offset (the buffer resource returns zeros)
#define GE2E_TF_LOAD_ROWS(REG, BI)                                                                       \
    do {                                                                                                 \
        int lq_ = lane;                                                                                  \
        asm volatile("" : "+v"(lq_));                                                                    \
        const unsigned vrow_ = 4 * lq_ < DG ? (unsigned)lq_ * 16u : OOB;                                 \
        const bool on_ = has_spk && (BI) < p.B;                                                          \
        const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(p.E + (size_t)(on_ ? (BI) : 0) * NM * DG, (unsigned)NM * ROWBG); \
        _Pragma("unroll") for (int i = 0; i < MR; ++i)                                                   \
            REG[i] = bload4<GE2E_T2_E_AUX>(rs_, (on_ && i < M) ? vrow_ : OOB, (unsigned)(j * M + min(i, M - 1)) * ROWBG); \
    } while (0)
// falling issue priority from barrier to barrier (the remedy of hazard 23 in the tiled contractions: the wave that is behind
// wins ties) measured -1 % here, as in the training kernel: off unless -DGE2E_TF_WITH_PRIO
#ifdef GE2E_TF_WITH_PRIO
#define GE2E_TF_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define GE2E_TF_PRIO(n)
#endif
// the same loads one row at a time (S issues them between its instruction groups): LOADR_SETUP once, then LOADR(REG, i)
#define GE2E_TF_LOADR_SETUP(BI)                                                                          \
    int lq_ = lane;                                                                                      \
    asm volatile("" : "+v"(lq_));                                                                        \
    const unsigned vrow_ = 4 * lq_ < DG ? (unsigned)lq_ * 16u : OOB;                                     \
    const bool on_ = has_spk && (BI) < p.B;                                                              \
    const __amdgpu_buffer_rsrc_t rs_ = make_rsrc(p.E + (size_t)(on_ ? (BI) : 0) * NM * DG, (unsigned)NM * ROWBG)
#define GE2E_TF_LOADR(REG, I_)                                                                           \
    do {                                                                                                 \
        if ((I_) < MR) {                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            REG[(I_) < MR ? (I_) : 0] = bload4<GE2E_T2_E_AUX>(rs_, (on_ && (I_) < M) ? vrow_ : OOB,      \
                                                              (unsigned)(j * M + min((I_), M - 1)) * ROWBG); \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    } while (0)
#define GE2E_TF_LANE()                                                  \
    int lv_ = lane;                                                     \
    asm volatile("" : "+v"(lv_));                                       \
    const int l15 = lv_ & 15, q = lv_ >> 4, d4 = 4 * lv_;               \
    const bool dact = D == 256 || d4 < D;                               \
    (void)l15; (void)q; (void)d4; (void)dact

    GE2E_PROF_DECL(20)
    {
        const int j0 = id.member * spm;
        const int my_spm = MEX ? 8 : max(0, min(spm, N - j0));
        const bool has_spk = wid < my_spm;
        const int j = j0 + wid;
        GE2E_TF_LOAD_ROWS(ra, id.team);
        GE2E_TF_LOAD_ROWS(rb2, id.team + id.nct);
    }
    const TeamId id_outer = id;
    bool failed = false;
    int nsig = 0;                       // signals this member has given (every member gives the same number)
    const int wid_outer = wid, member_outer = id.member, tid_outer = tid, m_outer = M;
    float4 sc_pend = zero4();           // member 0, wave 0: the eight members' scalars of batch sc_batch, requested, not yet summed
    int sc_batch = -1;
    auto sc_flush = [&]() __attribute__((always_inline)) {
        if (sc_batch >= 0) {            // (uniform: set by member 0's wave 0 only)
            const float l = oct_sum(sc_pend.x), a = oct_sum(sc_pend.y), c = oct_sum(sc_pend.z);
            if ((threadIdx.x & 63) == 0) {
                if (p.loss) p.loss[sc_batch] = l;
                if (p.dw) p.dw[sc_batch] = a;
                if (p.db) p.db[sc_batch] = c;
            }
            sc_batch = -1;
        }
    };
    // The body is written ONCE and expanded TWICE per loop trip, on alternating row-register sets: R0 holds the rows of the
    // batch being started and is reloaded, in place, with the rows of batch n + 2 as soon as A2 has consumed it; the other
    // set (batch n + 1, in flight) is not touched.  No register of an in-flight load is ever copied: with an explicit
    // "RA <- RB" the allocator sank the copies to the latch behind an s_waitcnt for the loads just issued, and `break`s in
    // the middle of the body made phis (and ~130 v_mov per trip) of the centroid fragments as well.  The trip behind the last
    // batch (seq == nb) runs the same body with have_cur false; exits only between two expansions.
    const int nb = id.team < p.B ? (p.B - id.team + id.nct - 1) / id.nct : 0;    // batches of this team
    int seq = 0;
    auto body = [&](float4 (&R0)[MR]) __attribute__((always_inline)) {
        // wave- and member-derived scalars are re-derived in every iteration from opaque copies (ge2e_team.hip, hazard 8)
        int wid_o = wid_outer, mem_o = member_outer, tid_o = tid_outer, m_o = m_outer;
        asm volatile("" : "+s"(wid_o), "+s"(mem_o), "+v"(tid_o), "+s"(m_o));
        const int wid = wid_o, tid = tid_o, lane = tid & 63, M = MEX ? MR : m_o, NM = N * M;
        TeamId id = id_outer;
        id.member = mem_o;
        const int j0 = id.member * spm;
        const int my_spm = MEX ? 8 : max(0, min(spm, N - j0));
        const bool has_spk = wid < my_spm;
        const int j = j0 + wid;
        const int kslot = 8 * id.member + wid;             // the slot this wave is responsible for
        const int rbase = wid * M;                         // first row of that speaker in the images
        const int tX = wid & 3, khX = wid >> 2;            // X: slot tile and K half of this wave
        const int bi = id.team + seq * id.nct;             // batch started in this iteration
        const bool have_cur = seq < nb, have_prev = seq > 0;
        const int buf = seq & 1, pbuf = buf ^ 1;
        sn_prev = sn_cur; ss_prev = ss_cur;

        // ===== requests: prev's centroid fragments (all members', complete since the poll at the end of the last iteration)
        // -> registers, for X(prev) below.  2 NCH loads of 1 KB per wave = 64 KB per workgroup through the CU's 64 B/clk
        // address path: issued in one go all eight waves stand at load issue for ~1 k cycles; they go out BETWEEN the
        // instruction groups of A1 instead, whose vector work covers them.
        h8 xa[NCH][2];          // slots 16 tX + l15, K-steps khX NCH + s, 8 q ..
        int lx_ = lane;
        asm volatile("" : "+v"(lx_));
        const unsigned oa = have_prev ? XO.chr[pbuf] + ((unsigned)tX * (4u * NCH) + (unsigned)khX * NCH) * 1024u + (unsigned)lx_ * 16u : OOB;
#define GE2E_TF_XA_LOAD(K_)                                                                                        \
    do {                                                                                                           \
        if ((K_) < 2 * NCH) {                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
            xa[((K_) < 2 * NCH ? (K_) : 0) >> 1][(K_) & 1] =                                                       \
                bload_h8<AUX_L2>(rsX, oa + 1024u * (unsigned)((K_) >> 1) + (((K_) & 1) ? 2048u * NCH : 0u), 0);    \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }                                                                                                          \
    } while (0)
        if (!have_cur) static_for<0, 2 * NCH>([&](auto kc) { GE2E_TF_XA_LOAD(decltype(kc)::value); });
        // ===== A1(cur): speaker sum -> unit centroid -> published (fragment-major) ==================================
        if (have_cur) {
            GE2E_TF_LANE();
            GE2E_TF_CONSTS();
            float4 s = zero4();
            static_for<0, MR>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (i < M) { s.x += R0[i].x; s.y += R0[i].y; s.z += R0[i].z; s.w += R0[i].w; }
                GE2E_TF_XA_LOAD(i);                        // (K_ >= 2 NCH: nothing)
            });
            const float4 c = scale4(s, inv_m);
            float sqs[2] = {dot4(c, c), dot4(s, s)};
            wave_sum_to_sgpr<2>(sqs);
            float rn, kap, nc;
            unit_stats_bf(sqs[0], eps_cos, eps_cos2, rn, kap, nc);
            if (!has_spk) { rn = 0.f; nc = 0.f; }              // slots without a speaker publish zero rows
            sn_cur = has_spk ? fM * nc : 0.f;
            ss_cur = has_spk ? sqs[1] : 0.f;
            h4 hi, lo;
            split4_scaled(c, rn * kSplitScale, hi, lo);
            // one 1-KB block per (slot tile, hi / lo, 32-column K-step) holding the 64 lanes' 16-byte MFMA fragments in lane
            // order (lane = 16 q + slot-in-tile): a consumer's load instruction reads 1 KB contiguously
            const unsigned blk = (unsigned)(kslot >> 4) * (4u * NCH) + (unsigned)(lv_ >> 3);
            const unsigned vh = dact ? blk * 1024u + (unsigned)(((lv_ >> 1) & 3) * 16 + (kslot & 15)) * 16u + (unsigned)(lv_ & 1) * 8u : OOB;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), rsX, vh + XO.chr[buf], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), rsX, vh + XO.chr[buf] + 2048u * NCH, 0, 0);
        }
        GE2E_PROF(0);
        GE2E_TF_PRIO(1);

        // ===== X(prev): X[slot][r] over this wave's K half -> LDS (fragments of the next K-step under the MFMAs) ======
        auto phase_x = [&]() __attribute__((always_inline)) {
        if (have_prev) {
            GE2E_TF_LANE();
            float* const XBk = khX ? XB1 : XB0;
            const int fx = (4 * (l15 & 3) + ((4 - (l15 >> 2)) & 3)) & ((D % 128 == 0) ? 15 : 7);   // et_off's f of rows 16 rb + l15
            h8 fb[2][2];
            f32x4 acc[2] = {acc_zero4(), acc_zero4()};
            // the swizzle depends on the row only through l15 (period 16): one lane offset per K-step, the row block and the
            // lo image are immediates (written per load, hipcc re-derived the XOR for each of the 20 fragments: 50 VALU)
            int xs[NCH];
#pragma unroll
            for (int s2 = 0; s2 < NCH; ++s2) xs[s2] = l15 * P + (((4 * (khX * NCH + s2) + q) ^ fx) << 3);
#define TF_X_LOAD(T_)                                                                                     \
    do {                                                                                                  \
        const _Float16* const pp_ = ETh + xs[(T_) % NCH] + 16 * ((T_) / NCH) * P;                         \
        fb[(T_) & 1][0] = frag_row(pp_);                                                                  \
        fb[(T_) & 1][1] = frag_row(pp_ + RT * P);                                                         \
    } while (0)
#define TF_X_STORE(RB_)                                                                       \
    *reinterpret_cast<float4*>(XBk + (16 * (RB_) + l15) * XP + 16 * tX + 4 * q) =             \
        make_float4(acc[(RB_) & 1][0], acc[(RB_) & 1][1], acc[(RB_) & 1][2], acc[(RB_) & 1][3])
            TF_X_LOAD(0);
#pragma unroll
            for (int rb = 0; rb < RBC; ++rb) {
                if (CT_X || rb < RBr) {
                    acc[rb & 1] = acc_zero4();
#pragma unroll
                    for (int s = 0; s < NCH; ++s) {
                        const int t = rb * NCH + s;
                        if (t + 1 < RBC * NCH && (CT_X || t + 1 < RBr * NCH)) TF_X_LOAD(t + 1);
                        mfma16x3(acc[rb & 1], xa[s][0], xa[s][1], fb[t & 1][0], fb[t & 1][1]);
                        __builtin_amdgcn_sched_barrier(0);   // fragments at most one K-step ahead (registers)
                    }
                    if (rb > 0) { TF_X_STORE(rb - 1); }
                }
            }
            if (CT_X) { TF_X_STORE(RBT - 1); }
            else {
#pragma unroll
                for (int rb = 0; rb < RBC; ++rb)
                    if (rb == RBr - 1) { TF_X_STORE(rb); }
            }
#undef TF_X_LOAD
#undef TF_X_STORE
        }
        };
        phase_x();
        sc_flush();
        GE2E_PROF(1);

        // ---- drain + BARRIER 1: cur's centroid is in L2 (and the member scalars of prev - 1); X(prev) is in LDS and the
        //      images are free.  Signalled in EVERY iteration: the one behind the last batch carries scalars only.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GE2E_PROF(2);
        __syncthreads();
        if (tid == 0) add_agent(&fl->c1, 1u);
        ++nsig;
        GE2E_PROF(3);
        GE2E_TF_PRIO(3);

        // ===== A2(cur), compute half: own rows -> |e|, e-hat as split-fp16 halves in registers; the row scalars go to their parity buffer
        h4 hiv[MR], lov[MR];
        auto phase_a2c = [&]() __attribute__((always_inline)) {
        if (have_cur && has_spk) {
            GE2E_TF_LANE();
            GE2E_TF_CONSTS();
            float eev[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) eev[i] = i < M ? dot4(R0[i], R0[i]) : 0.f;
            const float ee_l = wave_sums_scatter<MR>(eev, lv_);      // row i's |e|^2 in lane scatter_lane(i)
            float rne_l, ke_l, ne_l;
            unit_stats_bf(ee_l, eps_cos, eps_cos2, rne_l, ke_l, ne_l);
            {
                const int rho = lv_ >> 4, irow = 4 * (lv_ & 15) + (((rho & 1) << 1) | (rho >> 1));
                if ((lv_ & 15) < (MR + 3) / 4 && irow < M)
                    *reinterpret_cast<float4*>(RS + (buf * RT + rbase + irow) * 4) = make_float4(rne_l, ke_l, ee_l, ne_l);
            }
            const float rs_l = rne_l * kSplitScale;
            // all rows' hi halves, then all lo halves: a row's split is a chain of dependent mixed-precision FMAs; ten rows
            // side by side fill each other's latencies
            float scv[MR];
#pragma unroll
            for (int i = 0; i < MR; ++i) scv[i] = lane_get(rs_l, scatter_lane(i));
#pragma unroll
            for (int i = 0; i < MR; ++i)
                if (i < M) split4_scaled_hi_u(R0[i], scv[i], hiv[i]);
#pragma unroll
            for (int i = 0; i < MR; ++i)
                if (i < M) split4_scaled_lo_u(R0[i], scv[i], hiv[i], lov[i]);
        }
        };
        // (Tried: X(prev) -- matrix pipe -- and this compute half -- vector pipe -- in OPPOSITE order on the two waves of a SIMD,
        // in front of barrier 1: -6 %.  The same pairing of S(prev) with A2(cur): -5 %.  An MFMA in flight costs the partner
        // wave's vector stream ~6 cycles per instruction, tools/ubench: the two pipes of a SIMD do not run side by side.)
        phase_a2c();
        // ===== A2(cur), write half: e-hat halves -> images (X(prev) is done everywhere: barrier 1) ====================
        if (have_cur && has_spk) {
            GE2E_TF_LANE();
#pragma unroll
            for (int i = 0; i < MR; ++i)
                if (i < M && dact) {
                    const int eo = et_off<D>(rbase + i, d4);
                    *reinterpret_cast<h4*>(ETh + eo) = hiv[i];
                    *reinterpret_cast<h4*>(ETl + eo) = lov[i];
                }
        }
        // ---- the rows of batch n + 2 into the registers A2 has just finished with: they have until the next-but-one A1.
        // All eight waves asking for their 10 KB at the same point stand at load issue for ~1.3 k cycles (80 KB through the
        // CU's 64 B/clk address path); where there is an S(prev) to run, the requests go out one at a time BETWEEN its
        // instruction groups instead and the address path works underneath the vector work.
        const bool spread_loads = have_prev && has_spk;
        if (!spread_loads) GE2E_TF_LOAD_ROWS(R0, bi + 2 * id.nct);
        GE2E_PROF(4);
        GE2E_TF_PRIO(1);

        // ===== S(prev): leave-one-out statistics, softmax / contrast, per-row loss ==================================
        // Wave s = speaker slot s: FOUR lanes per row, 16 similarities per lane.  Lane (rr = lane >> 2, qq = lane & 3) holds
        // the slots (sb + j) & 63, j = 0..15, sb = (own slot & ~3) + 16 qq: aligned groups of four, and the own-speaker
        // column is always one of values 0..3 of the lane qq == 0.
        float loss_acc = 0.f, dw_acc = 0.f, db_acc = 0.f;
        if (have_prev && has_spk) {
            GE2E_TF_LANE();
            GE2E_TF_CONSTS();
            const float w2 = w * LOG2E, b2 = (w * eps + bias) * LOG2E, leps2 = __int_as_float(le_) * LOG2E;   // softmax in base 2
            const int rr = lv_ >> 2, qq = lv_ & 3;
            const bool rv = rr < M;
            const int r = rbase + min(rr, M - 1);
            const int ko = kslot;
            const int sb = (ko & ~3) + 16 * qq;
            const int jo = ko & 3;
            const bool own_lane = qq == 0;
            const bool all_valid = N == NC;
            const bool want_wb = p.dw != nullptr || p.db != nullptr;
            GE2E_TF_LOADR_SETUP(bi + 2 * id.nct);
            float x[16];
            float4 xa4[4], xb4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int s4 = (sb + 4 * jj) & (NC - 1);
                xa4[jj] = *reinterpret_cast<const float4*>(XB0 + r * XP + s4);
                xb4[jj] = *reinterpret_cast<const float4*>(XB1 + r * XP + s4);
            }
            GE2E_TF_LOADR(R0, 0);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 a = xa4[jj], b = xb4[jj];
                x[4 * jj + 0] = a.x + b.x; x[4 * jj + 1] = a.y + b.y; x[4 * jj + 2] = a.z + b.z; x[4 * jj + 3] = a.w + b.w;
                if (jj == 1) GE2E_TF_LOADR(R0, 1);
            }
            GE2E_TF_LOADR(R0, 2);
            const float xo = (XB0[r * XP + ko] + XB1[r * XP + ko]) * kSplitInv2;   // c-hat_j . e-hat_r
            const float4 rs0 = *reinterpret_cast<const float4*>(RS + (pbuf * RT + r) * 4);    // rne ke ee |e|
            const float rne = rs0.x, ee = rs0.z, ne = rs0.w;
            const float es = xo * sn_prev * ne;              // e . s_j
            const float eu = (es - ee) * inv_m1;
            const float uu = fmaxf((ss_prev - 2.0f * es + ee) * (inv_m1 * inv_m1), 0.0f);
            float rnu, ku, nu;
            unit_stats_bf(uu, eps_cos, eps_cos2, rnu, ku, nu);
            const float cosd = eu * rne * rnu;               // cos(e, leave-one-out centroid)
            GE2E_TF_LOADR(R0, 3);
            const float sjj2 = fmaf(w2, cosd, b2);
            const float w2s = w2 * kSplitInv2;               // S2 = w2s x + b2 on the raw accumulator sums
            auto vld = [&](int jx) {
                const int s = (sb + jx) & (NC - 1);
                return all_valid || (s & 7) < max(0, min(spm, N - (s >> 3) * spm));
            };
            bool ownj[4];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) ownj[jx] = own_lane && jo == jx;
            float per, coefsum = 0.f, db_row = 0.f;
            GE2E_TF_PRIO(0);
            GE2E_TF_LOADR(R0, 4);
            if (!CONTRAST) {
                float xm0 = ownj[0] ? x[1] : x[0], xm1 = ownj[1] ? x[0] : x[1], xm2 = ownj[2] ? x[3] : x[2], xm3 = ownj[3] ? x[2] : x[3];
                float xm;
                if (w2 >= 0.f) {
                    xm = fmaxf(fmaxf(xm0, xm1), fmaxf(xm2, xm3));
#pragma unroll
                    for (int jx = 4; jx < 16; jx += 2) xm = fmaxf(fmaxf(x[jx], x[jx + 1]), xm);
                } else {
                    xm = fminf(fminf(xm0, xm1), fminf(xm2, xm3));
#pragma unroll
                    for (int jx = 4; jx < 16; jx += 2) xm = fminf(fminf(x[jx], x[jx + 1]), xm);
                }
                GE2E_TF_LOADR(R0, 5);
                float mx = quad_max(fmaf(w2s, xm, b2));
                mx = fmaxf(fmaxf(mx, sjj2), leps2);
                const float t = b2 - mx;
                float gv[16];
#pragma unroll
                for (int jx = 0; jx < 16; ++jx) {
                    gv[jx] = __builtin_amdgcn_exp2f(fmaf(w2s, x[jx], t));
                    if ((jx & 3) == 3) GE2E_TF_LOADR(R0, 6 + (jx >> 2));
                }
                static_for<10, MR>([&](auto ic) { GE2E_TF_LOADR(R0, decltype(ic)::value); });
                if (!all_valid) {
#pragma unroll
                    for (int jx = 0; jx < 16; ++jx) gv[jx] = vld(jx) ? gv[jx] : 0.f;
                }
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) gv[jx] = ownj[jx] ? 0.f : gv[jx];
                // four partial sums side by side (one chain of 16 dependent adds is 16 x 8.9 cycles of latency)
                float z4[4] = {gv[0], gv[1], gv[2], gv[3]}, a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jx = 4; jx < 16; ++jx) z4[jx & 3] += gv[jx];
                const float zl = (z4[0] + z4[1]) + (z4[2] + z4[3]);
                float al = 0.f;
                if (want_wb) {
#pragma unroll
                    for (int jx = 0; jx < 16; ++jx) a4[jx & 3] = fmaf(gv[jx], x[jx], a4[jx & 3]);
                    al = (a4[0] + a4[1]) + (a4[2] + a4[3]);
                }
                const float zp = quad_sum(zl);                // sum over the other speakers, shifted
                const float zoff = zp + __builtin_amdgcn_exp2f(leps2 - mx);
                const float z = zoff + __builtin_amdgcn_exp2f(sjj2 - mx);
                per = LN2 * ((mx - sjj2) + __builtin_amdgcn_logf(z));
                if (want_wb) {
                    const float ap = quad_sum(al);
                    const float rz = rcp_nr(z);
                    const float ad0 = -zoff * rz;             // dL/dS on the own-speaker column: -(1 - p_jj)
                    coefsum = fmaf(ap * kSplitInv2, rz, ad0 * cosd);     // sum_k dL/dS_k c0_k
                    db_row = fmaf(zp, rz, ad0);
                }
            } else {
                static_for<5, MR>([&](auto ic) { GE2E_TF_LOADR(R0, decltype(ic)::value); });
                float best = -INFINITY, bx = 0.f; int besti = 0x7fffffff;
#pragma unroll
                for (int jx = 0; jx < 16; ++jx) {
                    const int s = (sb + jx) & (NC - 1);
                    const bool ok = vld(jx) && !(jx < 4 && ownj[jx & 3]);
                    const float sve = ok ? fmaf(w2s, x[jx], b2) : -INFINITY;
                    if (ok && (sve > best || (sve == best && s < besti))) { best = sve; besti = s; bx = x[jx]; }
                }
                const int loci = besti;
                quad_argmax(best, besti);
                bx = quad_sum(loci == besti && besti != 0x7fffffff ? bx : 0.f);
                const float pos = rcp_nr(1.0f + __builtin_amdgcn_exp2f(-sjj2));
                const float neg = (N > 1) ? rcp_nr(1.0f + __builtin_amdgcn_exp2f(-best)) : 0.0f;
                per = 1.0f - pos + neg;
                const float ad0 = -pos * (1.0f - pos);
                const float gn = neg * (1.0f - neg);
                coefsum = fmaf(gn, bx * kSplitInv2, ad0 * cosd);
                db_row = gn + ad0;
            }
            if (rv && own_lane) {
                loss_acc = per;
                dw_acc = fmaf(eps, db_row, coefsum);
                db_acc = db_row;
                if (p.per) p.per[(size_t)(bi - id.nct) * NM + j0 * M + rbase + rr] = per;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- member scalars: fixed-order reduction over the 8 waves ------------------------------------------------
        if (have_prev) {
            if (p.dw != nullptr || p.db != nullptr) {
                float v3[3] = {loss_acc, dw_acc, db_acc};
                wave_sum_to_sgpr<3>(v3);
                if (lane == 0) { RED[wid] = v3[0]; RED[8 + wid] = v3[1]; RED[16 + wid] = v3[2]; }
            } else {
                float v1[1] = {loss_acc};
                wave_sum_to_sgpr<1>(v1);
                if (lane == 0) { RED[wid] = v1[0]; RED[8 + wid] = 0.f; RED[16 + wid] = 0.f; }
            }
        }
        GE2E_PROF(5);

        // ===== W: cur's centroids of all members (signalled at barrier 1); BARRIER 2 ================================
        {
            int* const wsh = SH + 4 + (seq & 3);
            if (tid == 0) *wsh = spin_until(&fl->c1, (unsigned)(TEAM * nsig), ctl) ? 1 : 0;
            __syncthreads();                 // also: the images, row scalars and RED are written; XB has been read
            failed = *wsh == 0;              // (uniform; the loop ends below, behind requests that are harmless then)
        }
        GE2E_PROF(6);
        GE2E_TF_PRIO(3);

        // (cur's centroid fragments are requested at the top of the next iteration, between the instruction groups of its A1)
        // ---- member scalars of prev -> exchange (visible to member 0 after the NEXT signal); member 0: batch n - 2 out
        if (have_prev && tid == 0) {
            float l = 0.f, a = 0.f, c = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { l += RED[i]; a += RED[8 + i]; c += RED[16 + i]; }
            bstore4(rsX, SC4 + (unsigned)((seq - 1) & 3) * 128u + (unsigned)id.member * 16u, make_float4(l, a, c, 0.f));
        }
        // (requested here, summed and written a phase later -- sc_flush, behind the next X: member 0 must not stand in front of
        // an L2 round trip that the other seven members then wait for at the next hand-off)
        if (seq >= 2 && id.member == 0 && wid == 0) {
            GE2E_TF_LANE();
            sc_pend = bload4<AUX_L2>(rsX, lv_ < TEAM ? SC4 + (unsigned)((seq - 2) & 3) * 128u + (unsigned)lv_ * 16u : OOB, 0);
            sc_batch = bi - 2 * id.nct;
        }
        GE2E_PROF(7);
        GE2E_TF_PRIO(2);
        ++seq;
    };
    if (nb > 0) for (;;) {
        body(ra);
        if (seq > nb || failed) break;
        body(rb2);
        if (seq > nb || failed) break;
    }
    // ---- tail: the scalars of the last two batches.  The loop ended with iteration L (= this team's batch count): batch
    // L - 2 went to the exchange in iteration L - 1 and was covered by iteration L's signal; batch L - 1 went out just now.
    if (!failed) sc_flush();
    if (!failed) {
        int mem_o = member_outer, tid_o = tid_outer;
        asm volatile("" : "+s"(mem_o), "+v"(tid_o));
        const int tid = tid_o, lane = tid & 63;
        if (nb > 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) add_agent(&fl->c1, 1u);
            ++nsig;
            if (mem_o == 0 && tid < 64) {
                int okv = 1;
                if (lane == 0) okv = spin_until(&fl->c1, (unsigned)(TEAM * nsig), ctl) ? 1 : 0;
                if (__builtin_amdgcn_readfirstlane(okv) != 0) {
                    // iteration L read batch L - 2 (seq >= 2); a team with ONE batch has not read anything yet
                    const int first = nb >= 2 ? nb - 1 : 0;
                    for (int k = first; k < nb; ++k) {
                        const float4 scv = bload4<AUX_L2>(rsX, lane < TEAM ? SC4 + (unsigned)(k & 3) * 128u + (unsigned)lane * 16u : OOB, 0);
                        const float l = oct_sum(scv.x), a = oct_sum(scv.y), c = oct_sum(scv.z);
                        if (lane == 0) {
                            const int bo = id.team + k * id.nct;
                            if (p.loss) p.loss[bo] = l;
                            if (p.dw) p.dw[bo] = a;
                            if (p.db) p.db[bo] = c;
                        }
                    }
                } else {
                    failed = true;
                }
            }
        }
    }
    if (failed && (threadIdx.x & 63) == 0) __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gave_up = gave_up || failed;
    GE2E_PROF_FLUSH(20)
    }();
    // ---- end of the launch (team_finish, ge2e_team.hpp): one load and one atomic in the common case, the last workgroup hands
    //      the control block back clean; with the abort word up the workgroups that are still there redo the call
    {
        if (p.test_abort == 2 && blockIdx.x == (gridDim.x >> 1)) {   // diagnostics: the word rises in the middle of the grid's finish
            if (threadIdx.x == 0) __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
        }
        int* const fsh = reinterpret_cast<int*>(smem_f) + 4;
        const TeamRedo rd = team_finish(ctl, fsh, (int)(L.head_bytes / 16), gave_up);
        if (rd.n != 0) {
            team_redo<NCH>(p, L, F, smem_f, rd.n, rd.rank);
            team_redo_done(ctl, fsh, (int)(L.head_bytes / 16), rd.n);
        }
    }
#undef GE2E_TF_LOAD_ROWS
#undef GE2E_TF_LANE
#undef GE2E_TF_CONSTS
}

template <int NCH, int MR, int RBT, bool CONTRAST>
static hipError_t launch_fwd_nch(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream) {
    const void* fn = reinterpret_cast<const void*>(ge2e_team_fwd_kernel<NCH, MR, RBT, CONTRAST>);
    static KernelLaunchState state;
    unsigned lds = (unsigned)team_fwd_lds_bytes(L.rt, 64 * NCH);             // (NCH = ceil(D / 64): padded columns)
    if (fused_split_lds_bytes(p.D) > lds) lds = (unsigned)fused_split_lds_bytes(p.D);   // ... or the redo body's
    int nb = 0;
    hipError_t err = prepare_kernel(state, fn, 512, lds, &nb);
    if (err != hipSuccess) return err;
    if (p.test_abort == 1) {
        err = launch_team_head_init(p.ws, L.head_bytes, true, stream);
        if (err != hipSuccess) return err;
    }
    int grid = team_grid(p.B);
    if (p.grid_cap > 0 && p.grid_cap < grid)
        grid = p.grid_cap / (MAX_XCD * TEAM) * (MAX_XCD * TEAM) > 0 ? p.grid_cap / (MAX_XCD * TEAM) * (MAX_XCD * TEAM) : MAX_XCD * TEAM;
    if (nb < 1 || grid > nb * device_cu_count()) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL((ge2e_team_fwd_kernel<NCH, MR, RBT, CONTRAST>), dim3(grid), dim3(512), lds, stream, p, L, F);
    return hipGetLastError();
}
template <int NCH, int MR>
static hipError_t launch_fwd_variant(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream) {
    if (NCH == 4 && MR == 10 && L.rt == 80 && p.M == 10 && p.N == 64 && p.D == 256)     // the metric shape: compile-time N, M, D
        return p.variant == 1 ? launch_fwd_nch<4, 10, 5, true>(p, L, F, stream) : launch_fwd_nch<4, 10, 5, false>(p, L, F, stream);
    return p.variant == 1 ? launch_fwd_nch<NCH, MR, 0, true>(p, L, F, stream) : launch_fwd_nch<NCH, MR, 0, false>(p, L, F, stream);
}

// the team launch of a forward-only call (p.dE == NULL)
hipError_t launch_team_fwd(Problem& p, TeamKWs& L, const FusedWs& F, hipStream_t stream) {
    if (p.M <= 10) {
        switch ((p.D + 63) / 64) {
            case 1: return launch_fwd_variant<1, 10>(p, L, F, stream);
            case 2: return launch_fwd_variant<2, 10>(p, L, F, stream);
            case 3: return launch_fwd_variant<3, 10>(p, L, F, stream);
            default: return launch_fwd_variant<4, 10>(p, L, F, stream);
        }
    }
    switch ((p.D + 63) / 64) {
        case 1: return launch_fwd_variant<1, 16>(p, L, F, stream);
        case 2: return launch_fwd_variant<2, 16>(p, L, F, stream);
        case 3: return launch_fwd_variant<3, 16>(p, L, F, stream);
        default: return launch_fwd_variant<4, 16>(p, L, F, stream);
    }
}

}  // namespace ge2e
