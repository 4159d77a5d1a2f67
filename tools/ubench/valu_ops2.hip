// Issue cost of individual VALU instructions on gfx950 at 1, 2 and 4 waves per SIMD (tools only).
// 8 independent chains per wave, 32 instructions per loop iteration; cycles per instruction per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o valu_ops valu_ops.hip && ./valu_ops
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}


#define BODY(ASM)                                                                               \
    float x[8], y[8]; unsigned sg[8]; unsigned long long m2[8];                                 \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = x[i]; sg[i] = 0; m2[i] = 0; } \
    __syncthreads();                                                                            \
    const unsigned long long t0 = now();                                                        \
    for (int it = 0; it < iters; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ASM;                                  \
    }                                                                                           \
    const unsigned long long t1 = now();                                                        \
    float s = 0.f;                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) s += x[i] + y[i] + (float)sg[i] + (float)m2[i]; \
    if (s == 12345.678f) sink[0] = s;                                                           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;

template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters, float a, float b, unsigned long long msk, float sa) {
    if constexpr (OP == 0) { BODY(asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(a))) }
    if constexpr (OP == 1) { BODY(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 2) { BODY(asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 3) { BODY(asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 4) { BODY(asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(msk))) }
    if constexpr (OP == 5) { BODY(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg[i]) : "v"(x[i]))) }
    if constexpr (OP == 6) { BODY(asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(x[i]) : "s"(sa))) }
    if constexpr (OP == 7) { BODY(asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m2[i]) : "v"(x[i]), "v"(a))) }
    if constexpr (OP == 8) { BODY(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 9) { BODY(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(y[i]) : "v"(x[i]), "v"(a), "v"(b))) }
    if constexpr (OP == 10) { BODY(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 11) { BODY(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 12) { BODY(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 13) { BODY(asm volatile("v_min_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 14) { BODY(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(a))) }
    if constexpr (OP == 15) { BODY(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 16) { BODY(asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[i]) : "s"(sa))) }
    if constexpr (OP == 17) { BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "s"(sa), "v"(b))) }
    if constexpr (OP == 18) { BODY(asm volatile("v_log_f32 %0, %0" : "+v"(x[i]))) }
    if constexpr (OP == 19) { BODY(asm volatile("v_rsq_f32 %0, %0" : "+v"(x[i]))) }
    if constexpr (OP == 20) { BODY(asm volatile("v_permlane16_swap_b32 %0, %0" : "+v"(x[i]))) }
    if constexpr (OP == 21) { BODY(asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x[i]))) }
    if constexpr (OP == 22) { BODY(asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(x[i]))) }
    if constexpr (OP == 23) { BODY(asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 24) { BODY(asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(x[i]))) }
    if constexpr (OP == 25) { BODY(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 26) { BODY(asm volatile("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x[i]), "+v"(y[i]) : "v"(a), "v"(b))) }
    if constexpr (OP == 27) { BODY(asm volatile("v_max_f32 %0, %0, %2\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x[i]), "+v"(y[i]) : "v"(a), "v"(b))) }
}
static const char* NAMES[] = {"v_mov_b32", "v_add_u32", "v_lshl_add_u32", "v_and_b32", "v_cndmask_b32(sgpr)", "v_readlane_b32", "v_writelane_b32", "v_cmp_lt_f32(sgpr dst)", "v_fma_f32 x,x,a,a", "v_fma_f32 3 distinct+dst", "v_fmac_f32", "v_sub_f32", "v_max3_f32", "v_min_f32", "v_mul_lo_u32", "v_mad_u32_u24", "v_mul_f32 x,sgpr,x", "v_fma_f32 x,x,sgpr,v", "v_log_f32", "v_rsq_f32", "v_permlane16_swap", "v_add_f32_dpp row_ror", "v_add_f32_dpp row_bcast15", "v_pk_fma_f16", "v_cvt_f16_f32", "v_perm_b32", "v_exp_f32 + v_fma_f32 (per pair)", "v_max_f32 + v_fma_f32 (per pair)"};

template <int OP>
static void run(unsigned long long* out, float* sink, int cus, std::vector<unsigned long long>& h) {
    printf("%-42s", NAMES[OP]);
    for (int threads : {256, 512, 1024}) {
        double c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), 0, 0, out, sink, 1000, 1.0001f, 0.5f, 0x5555555555555555ull, 1.5f);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, (size_t)cus * 16 * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v;
            for (int b = 0; b < cus; ++b) { unsigned long long m = 0; for (int w = 0; w < threads / 64; ++w) m = std::max(m, h[(size_t)b * 16 + w]); v.push_back(m); }
            std::sort(v.begin(), v.end());
            c = (double)v[v.size() / 2];
        }
        printf("  %dw/SIMD: %5.2f", threads / 256, c / (1000.0 * 32 * (threads / 256)));
    }
    printf("\n");
}
template <int I, int N> struct Loop { static void go(unsigned long long* o, float* s, int c, std::vector<unsigned long long>& h) { run<I>(o, s, c, h); Loop<I + 1, N>::go(o, s, c, h); } };
template <int N> struct Loop<N, N> { static void go(unsigned long long*, float*, int, std::vector<unsigned long long>&) {} };
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, (size_t)cus * 16 * 8)); CK(hipMalloc(&sink, 256));
    std::vector<unsigned long long> h((size_t)cus * 16);
    printf("cycles per instruction per SIMD\n");
    Loop<0, 28>::go(out, sink, cus, h);
    return 0;
}
