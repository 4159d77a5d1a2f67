// Diagnostic entry points: exercise the device building blocks (cross-lane reductions, the
// split-fp16 tile contractions with their fragment / transposing-read layouts) on caller data so
// tests can check them against numpy in isolation.  Not used by the product path.
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_selftest.hpp"
#include "ge2e_team.hpp"

namespace ge2e {

namespace {
constexpr int KD = 256, PH = KD + 8, GP = 64 + 8;
}

// A [64][256], Bm [64][256], G [64][64] fp32 (|x| <= 1)  ->
//   X  [64][64]  = A . Bm^T          (gemm_nt, K contiguous in both images)
//   GE [64][256] = G . A             (gemm_nn, A read through the transposing load)
//   GC [64][256] = G^T . Bm          (gemm_tn, both read through the transposing load)
__global__ __launch_bounds__(256) void ge2e_selftest_split_kernel(const float* A, const float* Bm, const float* G,
                                                                  float* X, float* GE, float* GC) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    _Float16* Ahi = sm;
    _Float16* Alo = Ahi + 64 * PH;
    _Float16* Bhi = Alo + 64 * PH;
    _Float16* Blo = Bhi + 64 * PH;
    _Float16* Ghi = Blo + 64 * PH;
    _Float16* Glo = Ghi + 64 * GP;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 64 * KD / 4; i += 256) {
        const int r = i / (KD / 4), c = (i % (KD / 4)) * 4;
        h4 hi, lo;
        float4 a = reinterpret_cast<const float4*>(A)[i];
        split4(make_float4(a.x * kSplitScale, a.y * kSplitScale, a.z * kSplitScale, a.w * kSplitScale), hi, lo);
        *reinterpret_cast<h4*>(Ahi + r * PH + c) = hi;
        *reinterpret_cast<h4*>(Alo + r * PH + c) = lo;
        float4 b = reinterpret_cast<const float4*>(Bm)[i];
        split4(make_float4(b.x * kSplitScale, b.y * kSplitScale, b.z * kSplitScale, b.w * kSplitScale), hi, lo);
        *reinterpret_cast<h4*>(Bhi + r * PH + c) = hi;
        *reinterpret_cast<h4*>(Blo + r * PH + c) = lo;
    }
    for (int i = tid; i < 64 * 64 / 4; i += 256) {
        const int r = i / 16, c = (i % 16) * 4;
        h4 hi, lo;
        float4 g = reinterpret_cast<const float4*>(G)[i];
        split4(make_float4(g.x * kSplitScale, g.y * kSplitScale, g.z * kSplitScale, g.w * kSplitScale), hi, lo);
        *reinterpret_cast<h4*>(Ghi + r * GP + c) = hi;
        *reinterpret_cast<h4*>(Glo + r * GP + c) = lo;
    }
    __syncthreads();
    const int l31 = lane & 31, h = lane >> 5;
    {
        const int a = wid >> 1, b = wid & 1;
        f32x16 acc = gemm_nt_32x32<KD>(Ahi, Alo, PH, 32 * a, Bhi, Blo, PH, 32 * b, lane);
        for (int i = 0; i < 16; ++i) {
            const int m = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
            X[m * 64 + 32 * b + l31] = acc[i] * kSplitInv2;
        }
    }
    f32x16 o[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) o[a][b][i] = 0.f;
    gemm_nn_64x64(Ghi, Glo, GP, Ahi, Alo, PH, 64 * wid, lane, o);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) {
        const int m = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
        GE[m * KD + 64 * wid + 32 * b + l31] = o[a][b][i] * kSplitInv2;
        o[a][b][i] = 0.f;
    }
    gemm_tn_64x64(Ghi, Glo, GP, 0, Bhi, Blo, PH, 64 * wid, lane, o);
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) {
        const int m = 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h;
        GC[m * KD + 64 * wid + 32 * b + l31] = o[a][b][i] * kSplitInv2;
    }
}

// out[0..63] = wave_sum(x), out[64..127] = wave_max(x), out[128..191] = row16_sum, out[192..255] = quad_sum,
// out[256..319] = wave_argmax index, out[320..383] = quad_argmax index  (x: 64 floats, one wave)
__global__ void ge2e_selftest_wave_kernel(const float* x, float* out) {
    const int l = threadIdx.x;
    const float v = x[l];
    out[l] = wave_sum(v);
    out[64 + l] = wave_max(v);
    out[128 + l] = row16_sum(v);
    out[192 + l] = quad_sum(v);
    float bv = v; int bi = l;
    wave_argmax(bv, bi);
    out[256 + l] = (float)bi;
    bv = v; bi = l;
    quad_argmax(bv, bi);
    out[320 + l] = (float)bi;
}

// One wave: CH [64][256], R [16][256] fp32 (|x| <= 1, |CH . R| <= 1)  ->
//   XT [64][16]  = CH . R^T           (gemm_x_16rows: 16 x 16 x 32 tiles, d contiguous in both images)
//   GE [16][256] = XT^T . CH          (XT straight from the accumulators as the A operand, CH through the
//                                      transposing load: the chain the team kernel runs per speaker)
//   GT [16][256] = GE again, but written through the in-quad transpose (one row x 4 columns per lane)
__global__ __launch_bounds__(64) void ge2e_selftest_rows16_kernel(const float* CH, const float* R, float* XT,
                                                                  float* GE, float* GT) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    _Float16* Chi = sm;
    _Float16* Clo = Chi + 64 * PH;
    _Float16* Rhi = Clo + 64 * PH;
    _Float16* Rlo = Rhi + 16 * PH;
    const int lane = threadIdx.x;
    for (int i = lane; i < 80 * KD / 4; i += 64) {
        const int r = i / (KD / 4), c = (i % (KD / 4)) * 4;
        h4 hi, lo;
        const float4 a = r < 64 ? reinterpret_cast<const float4*>(CH)[i] : reinterpret_cast<const float4*>(R)[i - 64 * KD / 4];
        split4(make_float4(a.x * kSplitScale, a.y * kSplitScale, a.z * kSplitScale, a.w * kSplitScale), hi, lo);
        *reinterpret_cast<h4*>((r < 64 ? Chi + r * PH : Rhi + (r - 64) * PH) + c) = hi;
        *reinterpret_cast<h4*>((r < 64 ? Clo + r * PH : Rlo + (r - 64) * PH) + c) = lo;
    }
    __syncthreads();
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    gemm_x_16rows<KD>(Chi, Clo, PH, Rhi, Rlo, l15 * PH, lane, acc);
    f32x4 g[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 4; ++i) {
            const float x = acc[t][i] * kSplitInv2;
            XT[(16 * t + 4 * q + i) * 16 + l15] = x;
            g[t][i] = x * kSplitScale;
        }
    const GFrag gf = g_to_frag(g);
    for (int t = 0; t < 16; ++t) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        o = gemm_g_ch_tile(gf, Chi, Clo, PH, 16 * t, lane, o);
        for (int i = 0; i < 4; ++i) GE[(4 * q + i) * KD + 16 * t + l15] = o[i] * kSplitInv2;
        float x[4] = {o[0] * kSplitInv2, o[1] * kSplitInv2, o[2] * kSplitInv2, o[3] * kSplitInv2};
        quad_transpose4(x, lane);
        *reinterpret_cast<float4*>(GT + (4 * q + (lane & 3)) * KD + 16 * t + 4 * (l15 >> 2)) = make_float4(x[0], x[1], x[2], x[3]);
    }
}

// Team formation + the L2 hand-off protocol of ge2e_team.hpp under load: every member publishes
// `payload` float4 per round, all members read all eight payloads back and count mismatches.
// out[0] = complete teams, out[1] = mismatching float4, out[2..9] = workgroups per XCD, out[10] = abort word
__global__ __launch_bounds__(512) void ge2e_selftest_team_kernel(TeamCtl* ctl, TeamFlags* flags, float4* data,
                                                                  int rounds, int payload, unsigned* out) {
    __shared__ int sh[8];
    const int tid = threadIdx.x;
    const TeamId id = team_form(ctl, sh);
    if (blockIdx.x == 0 && tid < MAX_XCD) out[2 + tid] = ld_poll(&ctl->xcd_count[tid][0]);
    if (blockIdx.x == 0 && tid == 0) out[0] = (unsigned)id.nct;
    if (id.team < 0) return;
    TeamFlags* fl = flags + id.team;
    unsigned bad = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        data + (size_t)id.team * 2 * TEAM * payload, 0, (int)(2 * TEAM * payload * 16), 0x00020000);
    for (int r = 0; r < rounds; ++r) {
        const int buf = r & 1;
        float4* mine = data + ((size_t)(id.team * 2 + buf) * TEAM + id.member) * payload;
        for (int i = tid; i < payload; i += 512) {
            const float v = (float)(((id.team * 8 + id.member) * 131 + r) * 7 + i);
            mine[i] = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        }
        team_signal(&fl->c1);
        if (!team_wait(&fl->c1, (unsigned)(TEAM * (r + 1)), ctl, sh + 4)) break;
        for (int m = 0; m < TEAM; ++m)
            for (int i = tid; i < payload; i += 512) {
                const float v = (float)(((id.team * 8 + m) * 131 + r) * 7 + i);
                const unsigned off = (unsigned)((buf * TEAM + m) * payload + i) * 16u;
                const float4 got = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
                bad += (got.x != v) | (got.y != v + 1.f) | (got.z != v + 2.f) | (got.w != v + 3.f);
            }
        team_signal(&fl->c2);
        if (!team_wait(&fl->c2, (unsigned)(TEAM * (r + 1)), ctl, sh + 4)) break;
    }
    if (bad) add_agent(out + 1, bad);
    if (tid == 0 && ld_poll(&ctl->abort_)) out[10] = 1;
}

hipError_t launch_selftest_rows16(const float* CH, const float* R, float* XT, float* GE, float* GT, hipStream_t stream) {
    const size_t lds = (size_t)(2 * 80 * PH) * sizeof(_Float16);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_selftest_rows16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(ge2e_selftest_rows16_kernel, dim3(1), dim3(64), lds, stream, CH, R, XT, GE, GT);
    return hipGetLastError();
}

size_t selftest_team_bytes(int payload) {
    return sizeof(TeamCtl) + 64 * sizeof(TeamFlags) + (size_t)64 * 2 * TEAM * payload * 16;
}
hipError_t launch_selftest_team(void* ws, size_t ws_bytes, int grid, int rounds, int payload, unsigned* out,
                                hipStream_t stream) {
    if (grid < 1 || grid > 64 * TEAM || ws_bytes < selftest_team_bytes(payload)) return hipErrorInvalidValue;
    hipError_t err = launch_team_head_init(ws, (sizeof(TeamCtl) + 64 * sizeof(TeamFlags) + 15) / 16 * 16, false, stream);
    if (err != hipSuccess) return err;
    err = hipMemsetAsync(out, 0, 16 * sizeof(unsigned), stream);
    if (err != hipSuccess) return err;
    TeamCtl* ctl = reinterpret_cast<TeamCtl*>(ws);
    TeamFlags* flags = reinterpret_cast<TeamFlags*>(ctl + 1);
    float4* data = reinterpret_cast<float4*>(flags + 64);
    // all workgroups must be resident (see launch_nch in ge2e_team.hip): checked here, then an ordinary launch
    int per_cu = 0, dev = 0, cus = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(ge2e_selftest_team_kernel), 512, 0);
    if (err != hipSuccess) return err;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return hipErrorInvalidDevice;
    if (grid > per_cu * cus) return hipErrorCooperativeLaunchTooLarge;
    hipLaunchKernelGGL(ge2e_selftest_team_kernel, dim3(grid), dim3(512), 0, stream, ctl, flags, data, rounds, payload, out);
    return hipGetLastError();
}

hipError_t launch_selftest_split(const float* A, const float* Bm, const float* G, float* X, float* GE, float* GC,
                                 hipStream_t stream) {
    const size_t lds = (size_t)(4 * 64 * PH + 2 * 64 * GP) * sizeof(_Float16);
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(ge2e_selftest_split_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(ge2e_selftest_split_kernel, dim3(1), dim3(256), lds, stream, A, Bm, G, X, GE, GC);
    return hipGetLastError();
}

hipError_t launch_selftest_wave(const float* x, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(ge2e_selftest_wave_kernel, dim3(1), dim3(64), 0, stream, x, out);
    return hipGetLastError();
}

}  // namespace ge2e
