// Building blocks of GE2E_IMPL_TEAM: eight workgroups on eight CUs of ONE XCD share a batch.
//
//  * Team formation.  Every workgroup of the launch is resident (the host checks grid <= resident capacity, one per CU
//    because of its LDS footprint).  A workgroup reads the id of the XCD it landed on, takes a
//    ticket from that XCD's counter and waits until all workgroups of the grid have done so; the
//    final per-XCD counts then tell every workgroup the same story: tickets 8q..8q+7 of an XCD form
//    a complete team, leftover workgroups (count not a multiple of 8) exit.  Nothing relies on how
//    the dispatcher maps workgroup ids to XCDs.
//  * Hand-offs inside a team go through the XCD's own L2: the producer writes with plain stores,
//    every storing wave drains its stores (s_waitcnt vmcnt(0) = acknowledged by L2), the workgroup
//    barrier collects the waves and one lane bumps the team's counter with an agent-scope atomic.
//    Consumers poll the counter (sc1 load = L2-served) from one lane, pass a workgroup barrier and
//    read the bytes with sc1 loads only, which bypass the CU's L1 and therefore see the L2 line
//    the producer wrote.  Both sides sit behind the same L2, so no write-back / invalidate of L2
//    is needed -- that is the point of keeping a team inside one XCD.
//  * Every spin is bounded and watches a grid-wide abort word, so the grid always drains.
#pragma once
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include <cstddef>

namespace ge2e {

constexpr int TEAM = 8;            // workgroups (CUs) per batch
constexpr int MAX_XCD = 8;
// Every wait is bounded by TIME (s_memrealtime: the constant 100 MHz counter), not by a poll count: a hand-off between
// resident workgroups takes microseconds, so a member that has waited milliseconds is not going to be served -- another
// stream or process holds CUs the launch counted on -- and the sooner the abort word rises, the sooner the gated
// fall-back launch behind the kernel does the work.  (Round 2 counted 2^22 polls of s_sleep 2: seconds.)
constexpr unsigned long long TEAM_FORM_TICKS = 200000ull;      // 2 ms: every workgroup of the grid has started
constexpr unsigned long long TEAM_HANDOFF_TICKS = 400000ull;   // 4 ms: a hand-off inside a running team

// Control block at the head of the workspace.  SELF-CLEANING: a call leaves it the way it found it -- all zeros plus the
// magic word -- and ONE launch does everything (round 5; rounds 2-4 queued a gated one-workgroup-per-batch launch behind
// every team launch, which redid the call when the abort word was up and cleaned the block either way):
//   * every workgroup of the team launch ends in team_finish(): it counts itself in `done`, waits until the whole grid has
//     (all workgroups are resident -- the launch condition), reads the abort word and counts itself in `done`; the workgroup
//     that completes the count rewrites the block (nobody reads it any more).  With the abort word up -- no team formed, a
//     hand-off timed out -- ALL workgroups are still there and redo the call with the one-workgroup-per-batch body
//     (ge2e_fused_split_body.hpp), batches wg, wg + n, ...: never NaN, never stale outputs, no second launch;
//   * a block that does not carry the magic -- a fresh allocation, memory another implementation has written over -- holds
//     no counter that can be trusted: no teams, no end-of-grid wait; the workgroups redo the call with the same static
//     split and workgroup 0 writes a valid block when it is done.  ge2e_workspace_init() (include/ge2e_hip.h) writes a valid
//     block explicitly, so that a new workspace's first call already runs the team kernel; alloc_workspace does that.
// (Zeroed by a kernel, not a memset node: captured in a HIP graph beside torch's fill nodes, a memset node replayed with
// another node's pattern.)  Each word that is polled or bumped sits on its own 128-byte line.
constexpr unsigned TEAM_MAGIC = 0x6E2E7EA3u;
struct TeamCtl {
    unsigned arrived;   unsigned pad0[31];
    unsigned abort_;    unsigned pad1[31];
    unsigned nct;       unsigned pad2[31];              // complete teams (written by workgroup 0, diagnostic)
    unsigned xcd_count[MAX_XCD][32];                    // one line per XCD
    unsigned done;      unsigned pad3[31];              // team_finish, ONE word for both counts: low 16 bits = workgroups that have
                                                        // finished their batches, high 16 bits = those of them that found the abort
                                                        // word up and stay for the redo (a stayer adds 0x10001: its rank and its
                                                        // arrival are one atomic, so the add that completes the low half returns
                                                        // the final number of stayers)
    unsigned magic;                                     // TEAM_MAGIC <=> the rest of the block (and the team flags) is zero
    unsigned gen;       unsigned pad4[30];              // != 0: the launch (Problem::launch_seq) whose workgroup 0 wrote this block
                                                        // after a redo WITHOUT counters; that launch's own late workgroups must
                                                        // not take the block for a clean one (team_form).  Same 8 bytes as the
                                                        // magic: written by one store, read by one load
    unsigned fallbacks; unsigned pad5[31];              // diagnostic, survives the clean-up: calls on this workspace whose abort
                                                        // word was up (no team formed, a hand-off timed out, unclean block)
    unsigned go;        unsigned pad7[31];              // the redo's size, decided ONCE: 0 = not yet, TEAM_GO_LONE = every stayer by
                                                        // itself, else n = the number of stayers (stored by the last finisher)
    unsigned redo_done; unsigned pad8[31];              // workgroups of the redo that have finished it (team_redo_done)
};
static_assert(offsetof(TeamCtl, fallbacks) == 1664, "functional.workspace_fallback_count reads byte 1664");
static_assert(offsetof(TeamCtl, gen) == offsetof(TeamCtl, magic) + 4 && offsetof(TeamCtl, magic) % 16 == 0, "magic | gen: one 8-byte word");
constexpr unsigned TEAM_GO_LONE = 0xFFFFFFFFu;
constexpr unsigned long long TEAM_FINISH_TICKS = 5000000ull;    // 50 ms: the whole grid has finished (every inner wait is bounded
                                                                // by 4 ms and gives up as soon as the abort word rises)
// bytes of the control block + 64 per-team flag records (shape-independent; TeamKFlags = 3 lines, TeamFlags = 2)
constexpr size_t team_head_bytes() { return (sizeof(TeamCtl) + 64 * 3 * 128 + 255) / 256 * 256; }
// zero `bytes` at `head` (a multiple of 16) and set the magic (and, for diagnostics, the abort word): one small launch
hipError_t launch_team_head_init(void* head, size_t bytes, bool raise_abort, hipStream_t stream);
// a clean block (zeros + magic + the surviving fall-back count) over `n16` 16-byte pieces; called by one wave (64 threads).
// `gen` != 0 marks the block as written by that launch's redo without counters (TeamCtl::gen; it travels in the magic's piece).
__device__ __forceinline__ void team_head_rewrite(unsigned* head, int n16, unsigned fallbacks, unsigned gen = 0u) {
    constexpr int magic_piece = (int)(offsetof(TeamCtl, magic) / 16), fb_piece = (int)(offsetof(TeamCtl, fallbacks) / 16);
    uint4* h16 = reinterpret_cast<uint4*>(head);
    for (int i = threadIdx.x & 63; i < n16; i += 64)
        h16[i] = make_uint4(i == magic_piece ? TEAM_MAGIC : (i == fb_piece ? fallbacks : 0u), i == magic_piece ? gen : 0u, 0u, 0u);
}
struct TeamFlags {                                      // per team
    unsigned c1;        unsigned pad0[31];              // hand-off 1 (unit centroids published)
    unsigned c2;        unsigned pad1[31];              // hand-off 2 (partial centroid gradients published)
};

struct TeamId {
    int team;      // index among the complete teams, -1 = not in one, -2 = the control block cannot be trusted (no counters)
    int member;    // 0..7
    int nct;       // number of complete teams
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & (MAX_XCD - 1);
}
__device__ __forceinline__ unsigned ld_poll(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned add_agent(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane: wait until *p >= target.  Returns false when the launch is being aborted.
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, TeamCtl* ctl,
                                           unsigned long long ticks = TEAM_HANDOFF_TICKS) {
    if ((int)(ld_poll(p) - target) >= 0) return true;           // the common case: already there
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0;; ++it) {
        if ((int)(ld_poll(p) - target) >= 0) return true;
        if ((it & 63) == 63) {
            if (ld_poll(&ctl->abort_)) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(&ctl->abort_, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

// Called by all threads of the workgroup; `sh` is 4 ints of LDS.  Contains workgroup barriers.
// `launch_seq`: the host's number of this launch (0 = none): a block whose `gen` carries it was written by workgroup 0 of THIS
// launch at the end of a redo without counters -- to a workgroup of the same launch that starts late it is as untrusted as
// the block workgroup 0 found (it would otherwise form a team by itself, wait 2 ms, raise the abort word and wait 50 ms).
__device__ __forceinline__ TeamId team_form(TeamCtl* ctl, int* sh, unsigned launch_seq = 0u) {
    bool trusted = false;
    if (threadIdx.x == 0) {
        const unsigned long long mg = __hip_atomic_load(
            reinterpret_cast<const unsigned long long*>(__builtin_assume_aligned(&ctl->magic, 8)), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);      // magic | gen << 32, one load (the control block is 256-byte aligned, magic at 1536)
        trusted = (unsigned)mg == TEAM_MAGIC && (launch_seq == 0u || (unsigned)(mg >> 32) != launch_seq);
    }
    if (threadIdx.x == 0 && !trusted) {
        // not a clean control block (fresh or overwritten memory): no counter in it can be trusted, so no teams and no
        // end-of-grid protocol either -- the caller redoes the call with a static split and workgroup 0 writes a clean block
        sh[0] = -2; sh[1] = 0; sh[2] = 0;
    } else if (threadIdx.x == 0 && ld_poll(&ctl->abort_)) {   // launch already marked as failed (diagnostics): nobody forms a team
        sh[0] = -1; sh[1] = 0; sh[2] = 0;
    } else if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        const unsigned ticket = add_agent(&ctl->xcd_count[x][0], 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ticket add has been performed before "arrived" moves
        add_agent(&ctl->arrived, 1u);
        int team = -1, member = 0, nct = 0;
        if (spin_until(&ctl->arrived, gridDim.x, ctl, TEAM_FORM_TICKS)) {
            for (int i = 0; i < MAX_XCD; ++i) {
                const int full = (int)(ld_poll(&ctl->xcd_count[i][0]) / TEAM);
                if (i == (int)x && (int)ticket < full * TEAM) { team = nct + (int)ticket / TEAM; member = (int)ticket % TEAM; }
                nct += full;
            }
        }
        sh[0] = team; sh[1] = member; sh[2] = nct;
        if (blockIdx.x == 0) ctl->nct = (unsigned)nct;
    }
    __syncthreads();
    TeamId id;
    id.team = sh[0]; id.member = sh[1]; id.nct = sh[2];
    return id;
}

// End of a team launch on a TRUSTED control block, called by all threads of every workgroup (also the ones without a team or
// without a batch).  The common case costs one load and one atomic and waits for nobody: a workgroup that finds the abort
// word down counts itself in `done` and leaves; the one that completes the count rewrites the block (everybody else has
// left).  A workgroup that finds the abort word UP takes a rank and counts itself in `done` (one atomic) and STAYS until
// the last finisher has published how many of them there are (`go`).  Those that stayed
// redo the call with the one-workgroup-per-batch body, batches rank, rank + n, ...: all of the grid when no team could
// form (the word is up before anybody finishes), at least the team whose hand-off ran out otherwise.  Returns the number of
// workgroups in the redo (0: none, leave) and this workgroup's rank; the caller ends with team_redo_done().
// `sh`: 4 ints of LDS.  `n16`: size of the control block + flags in 16-byte pieces.
struct TeamRedo { int n, rank; };
__device__ __forceinline__ unsigned cas_agent(unsigned* p, unsigned expected, unsigned desired) {   // returns what was there
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
// The size of the redo is decided ONCE, by the last finisher, and published in `go` (round 6; until then every stayer read
// a counter `seen` for itself once `done` was complete -- but a last finisher that found the word risen late bumped `seen`
// AFTER that, and nothing ordered a stayer's two relaxed adds: stayers could leave with different n, and strides rank,
// rank + n, ... over different n do not tile the batches):
//   * `done` carries both counts: a workgroup that stays adds 0x10001 (its rank = the old high half), one that leaves adds 1.
//     One atomic per workgroup: there is no second add to order;
//   * the workgroup whose add completes the low half therefore holds the final number of stayers.  It stores it in `go` -- or,
//     with nobody staying, rewrites the block.  If it finds the abort word risen only now it does NOT join: whoever raised
//     the word stays, and the stayers cover every batch;
//   * stayers spin on `go`.  A stayer whose wait runs out (50 ms: a workgroup of this grid has not finished) moves `go` from
//     0 to TEAM_GO_LONE -- a compare-and-swap, so that either EVERY stayer redoes the call alone in the slice of its own block
//     index, or every stayer takes part in the ranked redo; the two never mix (they would share workspace slices).
// `known_up`: this workgroup knows the word is (being) raised -- no team formed anywhere, or its own hand-off ran out.
__device__ __forceinline__ TeamRedo team_finish(TeamCtl* ctl, int* sh, int n16, bool known_up) {
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this workgroup's results are out
        int n = 0, rank = 0, clean = 0;
        const unsigned up = ld_poll(&ctl->abort_) | (known_up ? 1u : 0u);
        const unsigned v = add_agent(&ctl->done, up ? 0x10001u : 1u);   // (grid <= 65 535 workgroups: one per CU)
        rank = (int)(v >> 16);
        const bool last = (v & 0xFFFFu) == gridDim.x - 1;
        unsigned g = 0u;
        if (last) {
            const unsigned s = (v >> 16) + (up ? 1u : 0u);          // final: every other workgroup has left or holds its rank
            if (s == 0u) clean = 1;                                 // nobody stays
            else {
                g = cas_agent(&ctl->go, 0u, s);                     // 0 -> n; a stayer may have declared TEAM_GO_LONE meanwhile
                if (g == 0u) g = s;
            }
        } else if (up) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned it = 0; (g = ld_poll(&ctl->go)) == 0u; ++it) {
                if ((it & 63) == 63 && __builtin_amdgcn_s_memrealtime() - t0 > TEAM_FINISH_TICKS) {
                    g = cas_agent(&ctl->go, 0u, TEAM_GO_LONE);
                    if (g == 0u) g = TEAM_GO_LONE;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (up && g == TEAM_GO_LONE) {
            // a workgroup of this grid has not finished within 50 ms: nothing about the block holds any more.  Take the
            // magic away (the next call redoes itself without counters and writes a fresh block) and redo ALONE.
            __hip_atomic_store(&ctl->magic, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            n = -1;
        } else if (up) {
            n = (int)g;
        }
        sh[0] = n; sh[1] = rank; sh[2] = clean;
    }
    __syncthreads();
    TeamRedo r{sh[0], sh[1]};
    if (sh[2] && threadIdx.x < 64)                                  // nobody else is left: a clean block for the next call
        team_head_rewrite(reinterpret_cast<unsigned*>(ctl), n16, ctl->fallbacks);
    return r;
}
// ... after the ranked redo: the last of its n workgroups rewrites the block (`redo_done` counts them; n is the same for all of
// them, so exactly one add returns n - 1 and `fallbacks` has a single writer)
__device__ __forceinline__ void team_redo_done(TeamCtl* ctl, int* sh, int n16, int n) {
    __syncthreads();
    if (n < 0) return;                                              // redone alone, the block is marked untrusted
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sh[0] = add_agent(&ctl->redo_done, 1u) == (unsigned)(n - 1) ? 1 : 0;
    }
    __syncthreads();
    if (sh[0] && threadIdx.x < 64) team_head_rewrite(reinterpret_cast<unsigned*>(ctl), n16, ctl->fallbacks + 1u);
}

// Producer side of a hand-off: call from ALL threads after the payload stores were issued.
__device__ __forceinline__ void team_signal(unsigned* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) add_agent(counter, 1u);
}
// Consumer side: all threads; returns false (uniformly) when aborted.  `sh` = 1 int of LDS.
__device__ __forceinline__ bool team_wait(const unsigned* counter, unsigned target, TeamCtl* ctl, int* sh) {
    if (threadIdx.x == 0) sh[0] = spin_until(counter, target, ctl) ? 1 : 0;
    __syncthreads();
    const bool ok = sh[0] != 0;
    __syncthreads();   // sh may be reused right away
    return ok;
}

// ---- 16 x 16 x 32 split-fp16 contractions (one wave) ---------------------------------------------
// v_mfma_f32_16x16x32_f16: A lane (m = lane & 15, kg = lane >> 4) supplies A[m][8 kg .. 8 kg + 7],
// B lane (n = lane & 15, kg) supplies B[8 kg .. 8 kg + 7][n]; C lane (n = lane & 15, q = lane >> 4)
// holds C[4 q + i][n] in register i.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma3_16(const h8& ah, const h8& al, const h8& bh, const h8& bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    return acc;
}

// X[k][r] for 64 centroid slots x 16 rows: acc[t][i] = sum_d CH[16 t + 4 q + i][d] * ROWS[row(l15)][d].
// Both images row-major with d contiguous; `row_off` = this lane's row offset (in halfs) into the row image.
template <int KD>
__device__ __forceinline__ void gemm_x_16rows(const _Float16* CHh, const _Float16* CHl, int pc,
                                              const _Float16* Rh, const _Float16* Rl, int row_off,
                                              int lane, f32x4 (&acc)[4]) {
    const int l15 = lane & 15, kg = lane >> 4;
    const int off_a = l15 * pc + 8 * kg;
    const int off_b = row_off + 8 * kg;
#pragma unroll
    for (int s = 0; s < KD / 32; ++s) {
        const h8 bh = frag_row(Rh + off_b + 32 * s), bl = frag_row(Rl + off_b + 32 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = mfma3_16(frag_row(CHh + off_a + 16 * t * pc + 32 * s), frag_row(CHl + off_a + 16 * t * pc + 32 * s),
                              bh, bl, acc[t]);
    }
}

// G (this wave's 16 rows x 64 centroid slots, in the accumulator layout of gemm_x_16rows, values
// already multiplied by kSplitScale) as the A operand of  gE[r][d] = sum_k G[r][k] CH[k][d]:
// K-step s covers the 32 centroid slots 32 s ..; slot e of lane group kg is k = 32 s + 4 kg + e
// (e < 4) and 32 s + 16 + 4 kg + e - 4 (e >= 4), i.e. registers g[2 s][0..3], g[2 s + 1][0..3].
struct GFrag { h8 hi[2], lo[2]; };
__device__ __forceinline__ GFrag g_to_frag(const f32x4 (&g)[4]) {
    GFrag f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        h4 h0, l0, h1, l1;
        split4(make_float4(g[2 * s][0], g[2 * s][1], g[2 * s][2], g[2 * s][3]), h0, l0);
        split4(make_float4(g[2 * s + 1][0], g[2 * s + 1][1], g[2 * s + 1][2], g[2 * s + 1][3]), h1, l1);
        f.hi[s] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        f.lo[s] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    return f;
}
// B operand of that contraction for the 16 columns cb.. of the row-major centroid image (K along rows):
// lane group kg reads block rows 32 s + 4 kg .. + 3 and 32 s + 16 + 4 kg .. + 3 through the transposing load.
__device__ __forceinline__ h8 frag_tr16(const _Float16* img, int pitch, int s, int cb, int lane) {
    const int kg = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const _Float16* p = img + (32 * s + 4 * kg + q) * pitch + cb + 4 * pp;
    const h4 t0 = tr_read4(p);
    const h4 t1 = tr_read4(p + 16 * pitch);
    return __builtin_shufflevector(t0, t1, 0, 1, 2, 3, 4, 5, 6, 7);
}
// acc[i] += gE[4 q + i][cb + l15]   (one 16 x 16 output tile)
__device__ __forceinline__ f32x4 gemm_g_ch_tile(const GFrag& g, const _Float16* CHh, const _Float16* CHl, int pc,
                                                int cb, int lane, f32x4 acc) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        acc = mfma3_16(g.hi[s], g.lo[s], frag_tr16(CHh, pc, s, cb, lane), frag_tr16(CHl, pc, s, cb, lane), acc);
    return acc;
}

}  // namespace ge2e
