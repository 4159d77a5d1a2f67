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

#define BODY(ASM, XT, INIT)                                                                     \
    XT x[8];                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) x[i] = INIT;                                  \
    __syncthreads();                                                                            \
    const unsigned long long t0 = now();                                                        \
    for (int it = 0; it < iters; ++it) {                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) ASM;                                  \
    }                                                                                           \
    const unsigned long long t1 = now();                                                        \
    float s = 0.f;                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) s += *reinterpret_cast<float*>(&x[i]);        \
    if (s == 12345.678f) sink[0] = s;                                                           \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;

template <int OP>
__global__ void k(unsigned long long* out, float* sink, int iters, float a, float b) {
    if constexpr (OP == 0) { BODY(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 1) { BODY(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 2) { f2 a2 = {a, b}; BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a2)), f2, (f2{threadIdx.x * 0.001f + i, 1.0f})) }
    if constexpr (OP == 3) { f2 a2 = {a, b}; BODY(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a2)), f2, (f2{threadIdx.x * 0.001f + i, 1.0f})) }
    if constexpr (OP == 4) { f2 a2 = {a, b}; BODY(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a2)), f2, (f2{threadIdx.x * 0.001f + i, 1.0f})) }
    if constexpr (OP == 5) { BODY(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 6) { BODY(asm volatile("v_exp_f32 %0, %0" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 7) { BODY(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 8) { BODY(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : ), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 9) { BODY(asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(x[i]) : "v"(a), "v"(b)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 10) { BODY(asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 11) { BODY(asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 12) { BODY(asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 13) { BODY(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a)), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 14) { BODY(asm volatile("v_rcp_f32 %0, %0" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
    if constexpr (OP == 15) { BODY(asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(x[i])), float, threadIdx.x * 0.001f + i) }
}
static const char* NAMES[] = {"v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cvt_pk_f16_f32", "v_exp_f32",
                              "v_add_f32_dpp(quad)", "v_cndmask_b32", "v_fma_mix_f32", "v_cvt_f32_f16", "v_mov_b32_dpp(row_shr)", "v_max_f32",
                              "v_mul_f32", "v_rcp_f32", "v_permlane32_swap"};
template <int OP>
static void run(unsigned long long* out, float* sink, int cus, std::vector<unsigned long long>& h) {
    printf("%-24s", NAMES[OP]);
    for (int threads : {256, 512, 1024}) {
        double c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(threads), 0, 0, out, sink, 1000, 1.0001f, 0.5f);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), out, (size_t)cus * 16 * 8, hipMemcpyDeviceToHost));
            std::vector<unsigned long long> v;
            for (int b = 0; b < cus; ++b) { unsigned long long m = 0; for (int w = 0; w < threads / 64; ++w) m = std::max(m, h[(size_t)b * 16 + w]); v.push_back(m); }
            std::sort(v.begin(), v.end());
            c = (double)v[v.size() / 2];
        }
        printf("  %dw/SIMD: %5.2f", threads / 256, c / (1000.0 * 32 * (threads / 256)));
    }
    printf("   (cycles per instruction per SIMD)\n");
}
int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    unsigned long long* out; float* sink;
    CK(hipMalloc(&out, (size_t)cus * 16 * 8)); CK(hipMalloc(&sink, 256));
    std::vector<unsigned long long> h((size_t)cus * 16);
    run<0>(out, sink, cus, h); run<1>(out, sink, cus, h); run<13>(out, sink, cus, h); run<12>(out, sink, cus, h);
    run<2>(out, sink, cus, h); run<3>(out, sink, cus, h); run<4>(out, sink, cus, h);
    run<5>(out, sink, cus, h); run<10>(out, sink, cus, h); run<9>(out, sink, cus, h);
    run<6>(out, sink, cus, h); run<14>(out, sink, cus, h);
    run<7>(out, sink, cus, h); run<11>(out, sink, cus, h); run<8>(out, sink, cus, h); run<15>(out, sink, cus, h);
    return 0;
}
