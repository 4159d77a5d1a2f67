// Shared device helpers for the GE2E HIP kernels (gfx950 / wave64 only).
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stddef.h>
#include <stdint.h>

namespace ge2e {

constexpr int kWave = 64;  // CDNA wavefront; hard-coded, this library targets gfx950 only

// Launch-time problem description shared by every kernel.
struct Problem {
    const float* E;   // [B][N][M][D]
    const float* w;   // device scalar (s3:16); null -> w_imm
    const float* b;   // device scalar (s3:17); null -> b_imm
    float w_imm, b_imm;
    float* loss;      // [B]
    float* per;       // [B][N][M] or null
    float* dE;        // [B][N][M][D] or null (forward only)
    float* dw;        // [B] or null
    float* db;        // [B] or null
    float* cos_out;   // [B][N][M][N] or null (ge2e_cos_sim)
    float* ws;        // workspace
    int B, N, M, D;
    int variant;
    float eps_cos;    // cosine_similarity eps (1e-8)
    float eps;        // hp.general.small_err (1e-6)
    float log_eps;    // logf(eps), -inf when eps == 0
    unsigned long long* prof;  // diagnostic builds (-DGE2E_PROFILE) only: per-phase cycle sums
    int grid_cap;              // > 0: at most this many workgroups (diagnostics: the selftest launch, max_workgroups)
    int test_abort;            // diagnostics: 1 = the team launch starts with its abort word raised (every workgroup stays for the
                               // in-launch redo); 2 = ONE workgroup raises it when it reaches the end of the launch (some leave, some stay)
    unsigned launch_seq;       // team launches: the host's number of this launch (never 0), see TeamCtl::gen
    // ge2e_loss_fwd_bwd_raw: E is the encoder's RAW projection Y [B][N*M][D] in its own (permuted) row order and
    // src [B][N*M] (or null = identity) says which row of Y is row r of the (N,M,D) block: the kernel normalises and gathers
    // in its load stage and writes dL/dY (through the normalisation's backward, scattered back) in its store stage.
    int raw;
    const int* src;
};

// In-kernel phase stamps (cdna_hip_programming.md section 7): compiled in only with
// -DGE2E_PROFILE (tools/profile_phases.py builds that variant); the shipped library has none.
#ifdef GE2E_PROFILE
#define GE2E_PROF_DECL(n)                                              \
    unsigned long long prof_acc[n];                                    \
    for (int prof_i = 0; prof_i < n; ++prof_i) prof_acc[prof_i] = 0;   \
    unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#define GE2E_PROF(i)                                                   \
    do {                                                               \
        __builtin_amdgcn_sched_barrier(0);                             \
        unsigned long long prof_now = __builtin_amdgcn_s_memtime();    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                            \
        prof_acc[i] += prof_now - prof_last;                           \
        prof_last = prof_now;                                          \
        __builtin_amdgcn_sched_barrier(0);                             \
    } while (0)
#ifndef GE2E_PROF_TID
#define GE2E_PROF_TID 0   /* whose view: the first lane of wave GE2E_PROF_TID / 64 */
#endif
#define GE2E_PROF_FLUSH(n)                                             \
    if (threadIdx.x == GE2E_PROF_TID && p.prof)                                    \
        for (int prof_i = 0; prof_i < n; ++prof_i) atomicAdd(p.prof + prof_i, prof_acc[prof_i]);
#define GE2E_PROF_FLUSH_AT(base, n)                                    \
    if (threadIdx.x == GE2E_PROF_TID && p.prof)                                    \
        for (int prof_i = 0; prof_i < n; ++prof_i) atomicAdd(p.prof + (base) + prof_i, prof_acc[prof_i]);
/* a stamp taken inside a callee (`t`: its s_memtime) closes phase i */
#define GE2E_PROF_AT(i, t)                                             \
    do {                                                               \
        prof_acc[i] += (t) - prof_last;                                \
        prof_last = (t);                                               \
    } while (0)
#define GE2E_PROF_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#elif defined(GE2E_MARKS)   // ISA reading aid: phase boundaries as comments in the -save-temps assembly, no code
#define GE2E_PROF_DECL(n)
#define GE2E_PROF(i) asm volatile("; PHASEMARK " #i)
#define GE2E_PROF_FLUSH(n)
#define GE2E_PROF_FLUSH_AT(base, n)
#define GE2E_PROF_AT(i, t)
#define GE2E_PROF_DRAIN()
#else
#define GE2E_PROF_DECL(n)
#define GE2E_PROF(i)
#define GE2E_PROF_FLUSH(n)
#define GE2E_PROF_FLUSH_AT(base, n)
#define GE2E_PROF_AT(i, t)
#define GE2E_PROF_DRAIN()
#endif

// ---- cross-lane reductions on the VALU (DPP + gfx950 permlane swaps), no LDS round trips ----
// hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-crossbar op with ~100-cycle dependent
// latency per step); these stay in the vector pipe.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each group of 8
constexpr int DPP_MIRROR = 0x140;      // lane i <-> 15 - i inside each row of 16

// lanes l and l^16 (SWAP16) / l and l^32 (SWAP32): both values of the pair, in either order
#define GE2E_SWAP16(u) __builtin_amdgcn_permlane16_swap((u), (u), false, false)
#define GE2E_SWAP32(u) __builtin_amdgcn_permlane32_swap((u), (u), false, false)

// 4 x 4 transpose inside each quad of lanes: lane p of the quad enters with x[q] = R[q][p] (four
// registers = four rows, its own column) and leaves with x[k] = R[p][k] (one row, four consecutive
// columns) -- two DPP butterfly stages, no LDS.
__device__ __forceinline__ void quad_transpose4(float (&x)[4], int lane) {
    // every DPP move is executed by ALL lanes before the selects: inside a ?: arm the compiler would
    // run it under a partial EXEC mask and the disabled source lanes would read as zero
    const bool even = (lane & 1) == 0, lo = (lane & 2) == 0;
    const float d0 = dpp_f<DPP_XOR1>(x[0]), d1 = dpp_f<DPP_XOR1>(x[1]);
    const float d2 = dpp_f<DPP_XOR1>(x[2]), d3 = dpp_f<DPP_XOR1>(x[3]);
    const float n0 = even ? x[0] : d1, n1 = even ? d0 : x[1];
    const float n2 = even ? x[2] : d3, n3 = even ? d2 : x[3];
    const float q0 = dpp_f<DPP_XOR2>(n0), q1 = dpp_f<DPP_XOR2>(n1);
    const float q2 = dpp_f<DPP_XOR2>(n2), q3 = dpp_f<DPP_XOR2>(n3);
    x[0] = lo ? n0 : q2;
    x[1] = lo ? n1 : q3;
    x[2] = lo ? q0 : n2;
    x[3] = lo ? q1 : n3;
}
__device__ __forceinline__ float quad_sum(float v) { v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); return v; }
__device__ __forceinline__ float quad_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v)); return v;
}
// sum / max over each aligned group of 8 lanes, result in all 8
__device__ __forceinline__ float oct_sum(float v) { v = quad_sum(v); v += dpp_f<DPP_HALF_MIRROR>(v); return v; }
__device__ __forceinline__ float oct_max(float v) { v = quad_max(v); return fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); }
// sum / max over each aligned group of 16 lanes, result in all 16
__device__ __forceinline__ float row16_sum(float v) {
    v = quad_sum(v); v += dpp_f<DPP_HALF_MIRROR>(v); v += dpp_f<DPP_MIRROR>(v); return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = quad_max(v); v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_MIRROR>(v)); return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    auto a = GE2E_SWAP16(__float_as_uint(v));
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = GE2E_SWAP32(__float_as_uint(v));
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// K wave sums at once, stage by stage: a single DPP reduction is a chain of dependent VALU -> DPP steps with two wait
// states each (hipcc fills them with s_nop: 9 per sum); K independent chains fill each other's slots.
template <int K>
__device__ __forceinline__ void wave_sum_n(float (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR2>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_HALF_MIRROR>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_MIRROR>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        auto a = GE2E_SWAP16(__float_as_uint(v[k]));
        v[k] = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        auto b = GE2E_SWAP32(__float_as_uint(v[k]));
        v[k] = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
}
// Sum over the wave into an SGPR: four DPP steps inside each row of 16 lanes, two row broadcasts (lane 15 of a row into
// the next row, lane 31 into rows 2 and 3) and one v_readlane of lane 63 -- 7 instructions a value against 12 for the
// all-lanes form, and the result is a scalar operand.  K values stage by stage (see wave_sum_n).
template <int K>
__device__ __forceinline__ void wave_sum_to_sgpr(float (&v)[K]) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR2>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_HALF_MIRROR>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_MIRROR>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k)   // row_bcast:15, rows 1 and 3
        v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x142, 0xA, 0xF, false));
#pragma unroll
    for (int k = 0; k < K; ++k)   // row_bcast:31, rows 2 and 3
        v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), 0x143, 0xC, 0xF, false));
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}
// K wave sums, "reduce-scatter" form: four DPP steps inside each row of 16 lanes, then the four rows are combined with
// half / row SWAPS that reduce TWO values per instruction (v_permlane32_swap puts value a's two half-sums into lanes 0-31
// and value b's into lanes 32-63; v_permlane16_swap does the same one level down): 4 K + ~1.6 K instructions against
// 12 K for wave_sum_to_sgpr + a v_writelane per value.  Result: ONE register in which lane scatter_lane(i) (and the rest
// of its row-position) holds the sum of v[i]; K <= 16.
constexpr int scatter_lane(int i) { return 16 * ((((i & 3) & 1) << 1) | ((i & 3) >> 1)) + (i >> 2); }   // row [0,2,1,3][i & 3], lane i / 4
template <int K>
__device__ __forceinline__ float wave_sums_scatter(float (&v)[K], int lane) {
    static_assert(K >= 1 && K <= 16, "at most four result registers of four values");
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR1>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_XOR2>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_HALF_MIRROR>(v[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += dpp_f<DPP_MIRROR>(v[k]);
    constexpr int K2 = (K + 1) / 2, K4 = (K2 + 1) / 2;
    float t[K2], u[K4];
#pragma unroll
    for (int p = 0; p < K2; ++p) {       // lanes 0-31: value 2 p (rows 0+2, 1+3), lanes 32-63: value 2 p + 1
        const unsigned a = __float_as_uint(v[2 * p]), b = __float_as_uint(v[2 * p + 1 < K ? 2 * p + 1 : 2 * p]);
        auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        t[p] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int m = 0; m < K4; ++m) {       // row 0: value 4 m, row 1: 4 m + 2, row 2: 4 m + 1, row 3: 4 m + 3
        const unsigned c = __float_as_uint(t[2 * m]), d = __float_as_uint(t[2 * m + 1 < K2 ? 2 * m + 1 : 2 * m]);
        auto sw = __builtin_amdgcn_permlane16_swap(c, d, false, false);
        u[m] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    float r = u[0];
    const int l15 = lane & 15;
#pragma unroll
    for (int m = 1; m < K4; ++m) r = l15 == m ? u[m] : r;
    return r;
}
template <int R, int RMAX, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (R < RMAX) {
        f(std::integral_constant<int, R>{});
        static_for<R + 1, RMAX>(f);
    }
}
template <int LANE>
__device__ __forceinline__ float lane_put(float vec, float uniform) {   // vec[LANE] = uniform (a wave-uniform value)
    const int u = __builtin_amdgcn_readfirstlane(__float_as_int(uniform));
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(u), "n"(LANE));
    return vec;
}
__device__ __forceinline__ float lane_get(float vec, int lane_id) {                  // uniform = vec[lane_id]
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vec), lane_id));
}

__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    auto a = GE2E_SWAP16(__float_as_uint(v));
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = GE2E_SWAP32(__float_as_uint(v));
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// (value, index) arg-max; ties resolve to the lowest index (torch.max picks the first).
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void quad_argmax(float& v, int& i) {
    argmax_merge(v, i, dpp_f<DPP_XOR1>(v), dpp_i<DPP_XOR1>(i));
    argmax_merge(v, i, dpp_f<DPP_XOR2>(v), dpp_i<DPP_XOR2>(i));
}
__device__ __forceinline__ void oct_argmax(float& v, int& i) {
    quad_argmax(v, i);
    argmax_merge(v, i, dpp_f<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(i));
}
__device__ __forceinline__ void row16_argmax(float& v, int& i) {
    quad_argmax(v, i);
    argmax_merge(v, i, dpp_f<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(i));
    argmax_merge(v, i, dpp_f<DPP_MIRROR>(v), dpp_i<DPP_MIRROR>(i));
}
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
    quad_argmax(v, i);
    argmax_merge(v, i, dpp_f<DPP_HALF_MIRROR>(v), dpp_i<DPP_HALF_MIRROR>(i));
    argmax_merge(v, i, dpp_f<DPP_MIRROR>(v), dpp_i<DPP_MIRROR>(i));
    {
        auto a = GE2E_SWAP16(__float_as_uint(v));
        auto b = GE2E_SWAP16((unsigned)i);
        float v0 = __uint_as_float(a[0]); int i0 = (int)b[0];
        argmax_merge(v0, i0, __uint_as_float(a[1]), (int)b[1]);
        v = v0; i = i0;
    }
    {
        auto a = GE2E_SWAP32(__float_as_uint(v));
        auto b = GE2E_SWAP32((unsigned)i);
        float v0 = __uint_as_float(a[0]); int i0 = (int)b[0];
        argmax_merge(v0, i0, __uint_as_float(a[1]), (int)b[1]);
        v = v0; i = i0;
    }
}

// x / max(|x|, eps) bookkeeping shared by every kernel: from a squared norm give
// the reciprocal of the clamped norm and kappa = clamped / true norm (0 for a zero
// vector).  ATen clamps the norm in place under no_grad, so the backward of
// x_hat = x / n_c is (g - kappa * (g . x_hat) * x_hat) / n_c  (oracle/_unit_bwd).
__device__ __forceinline__ void unit_stats(float sq, float eps_cos, float& rn, float& kappa) {
    float n = sqrtf(sq);
    float nc = fmaxf(n, eps_cos);
    rn = 1.0f / nc;
    kappa = n > 0.0f ? nc / n : 0.0f;
}

__host__ __device__ inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Compute units of the CURRENT device.  Looked up once per device and process (a process may drive several devices, and
// a partitioned MI355X shows fewer CUs): hipGetDevice is a thread-local read, the attribute query is not, and a launch
// asks several times.  Without a visible device (build host) size queries answer for a whole MI355X.
constexpr int kMaxDevices = 64;
inline int current_device_slot() {
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices) ? dev : -1;
}
inline int device_cu_count() {
    static int cache[kMaxDevices] = {};          // 0 = not asked yet; racing writers store the same value
    const int dev = current_device_slot();
    if (dev >= 0 && cache[dev] > 0) return cache[dev];
    int d = 0, n = 0;
    if (hipGetDevice(&d) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) {
        if (dev >= 0) cache[dev] = n;
        return n;
    }
    return 256;
}
// Per-(device, kernel) launch state of a kernel with dynamic LDS (two HIP API calls per launch otherwise).  Host threads may
// launch the same instantiation with different LDS sizes at once, so the state is guarded by a mutex and:
//  * the opt-in attribute (MaxDynamicSharedMemorySize) is only ever RAISED -- lowering it between another thread's
//    prepare_kernel and its launch would fail that launch;
//  * the occupancy is remembered per (device, LDS size) in a small table, never paired with another size's answer (the
//    team launch's co-residency check, grid <= blocks x CUs, depends on it).
struct KernelLaunchState {
    static constexpr int kSizes = 4;
    std::mutex mu;
    unsigned attr[kMaxDevices] = {};              // bytes the attribute has been raised to on this device
    unsigned lds[kMaxDevices][kSizes] = {};       // LDS sizes asked about ...
    int blocks[kMaxDevices][kSizes] = {};         // ... and hipOccupancyMaxActiveBlocksPerMultiprocessor at each
    int next[kMaxDevices] = {};                   // round-robin replacement
};
inline hipError_t prepare_kernel(KernelLaunchState& st, const void* fn, int threads, unsigned lds_bytes, int* blocks_per_cu) {
    const int dev = current_device_slot();
    std::lock_guard<std::mutex> lock(st.mu);
    if (dev >= 0 && st.attr[dev] >= lds_bytes)
        for (int i = 0; i < KernelLaunchState::kSizes; ++i)
            if (st.lds[dev][i] == lds_bytes && st.blocks[dev][i] > 0) {
                if (blocks_per_cu) *blocks_per_cu = st.blocks[dev][i];
                return hipSuccess;
            }
    hipError_t err;
    if (dev < 0 || st.attr[dev] < lds_bytes) {
        err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return err;
        if (dev >= 0) st.attr[dev] = lds_bytes;
    }
    int nb = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds_bytes);
    if (err != hipSuccess) return err;
    if (dev >= 0 && nb > 0) {
        const int i = st.next[dev];
        st.next[dev] = (i + 1) % KernelLaunchState::kSizes;
        st.lds[dev][i] = lds_bytes;
        st.blocks[dev][i] = nb;
    }
    if (blocks_per_cu) *blocks_per_cu = nb;
    return hipSuccess;
}

}  // namespace ge2e
