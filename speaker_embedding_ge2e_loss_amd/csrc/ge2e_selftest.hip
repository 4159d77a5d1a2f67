// Diagnostic entry points: exercise the device building blocks (cross-lane reductions, the
// split-fp16 tile contractions with their fragment / transposing-read layouts) on caller data so
// tests can check them against numpy in isolation.  Not used by the product path.
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
#include "ge2e_selftest.hpp"

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
