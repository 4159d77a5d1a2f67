// Per-CU rate calibration for the team kernel's budget (tools only, not part of the product):
//   valu    wave64 v_fma_f32 issue cost per SIMD at 1, 2, 4 waves per SIMD, independent and dependent chains
//   store   whole-line 16-byte nt stores / one-dword stores: cycles per instruction with 1..8 (16) waves of a CU storing
//   lds     ds_read_b128 / ds_read_b64_tr_b16 cycles per instruction at 1, 2, 4 waves per SIMD
//   mix     one wave of a SIMD issuing MFMAs beside a partner issuing VALU: what each pays
// One workgroup per CU (grid = CUs), cycles from s_memtime around the loop, median over workgroups printed.
//   hipcc -O3 --offload-arch=gfx950 -o cu_rates cu_rates.hip && ./cu_rates
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// ---- VALU ---------------------------------------------------------------------------------------
template <int CHAINS>
__global__ void k_valu(unsigned long long* out, float* sink, int iters, float a) {
    float x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / CHAINS; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
    }
    const unsigned long long t1 = now();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- stores -------------------------------------------------------------------------------------
// MODE 0: 16 bytes per lane, whole 128-byte lines (8 lanes per line), nt.  MODE 1: one dword per lane (256 B contiguous).
// MODE 2: 16 bytes per lane, default policy.  Only waves < nstore store; the others wait at the barrier.
template <int MODE>
__global__ void k_store(unsigned long long* out, float* dst, int nstore, int per_wave, size_t wg_stride_bytes) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(dst) + (size_t)blockIdx.x * wg_stride_bytes, 0, (int)wg_stride_bytes, 0x00020000);
    const u32x4 v = {1u, 2u, 3u, (unsigned)threadIdx.x};
    __syncthreads();
    const unsigned long long t0 = now();
    if (wid < nstore) {
        for (int i = 0; i < per_wave; ++i) {
            if (MODE == 1) {
                const unsigned off = (unsigned)((wid * per_wave + i) * 256 + lane * 4);
                __builtin_amdgcn_raw_buffer_store_b32(v.x, rs, off, 0, 0);
            } else {
                const unsigned off = (unsigned)((wid * per_wave + i) * 1024 + lane * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, MODE == 0 ? 2 : 0);
            }
        }
    }
    const unsigned long long t1 = now();   // issue done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = now();   // acknowledged
    if (lane == 0) { out[blockIdx.x * 32 + wid] = t1 - t0; out[blockIdx.x * 32 + 16 + wid] = t2 - t0; }
}

// ---- LDS reads ----------------------------------------------------------------------------------
template <int MODE>   // 0: ds_read_b128 (conflict-free rows 544 B apart), 1: ds_read_b64_tr_b16, 2: ds_read_b64
__global__ void k_lds(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l15 = lane & 15, q = lane >> 4;
    const char* p = smem + l15 * 544 + q * 16 + (threadIdx.x >> 6) * 64;
    if (MODE == 1) p = smem + ((lane & 15) >> 2) * 544 + (lane & 3) * 8 + q * 4 * 544 + (threadIdx.x >> 6) * 64;
    float acc = 0.f;
    const unsigned long long t0 = now();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "i"(u * 32 * 17));
                asm volatile("" :: "v"(v));
            } else if (MODE == 1) {
                h4 v;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "i"(u * 32));
                asm volatile("" :: "v"(v));
            } else {
                h4 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(smem + lane * 8 + (threadIdx.x >> 6) * 512)), "i"(u * 32));
                asm volatile("" :: "v"(v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = now();
    if (acc == 12345.f) sink[0] = acc;
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// ---- MFMA beside VALU ---------------------------------------------------------------------------
// waves < 4 (one per SIMD): MFMA chains; waves >= 4: VALU (if valu_on).  512 threads.
__global__ void k_mix(unsigned long long* out, float* sink, int iters, int mfma_on, int valu_on, float a) {
    const int wid = threadIdx.x >> 6;
    h8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (threadIdx.x & 7) + i); fb[i] = (_Float16)(0.02f * i); }
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    __syncthreads();
    const unsigned long long t0 = now();
    if (wid < 4) {
        if (mfma_on)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc1, 0, 0, 0);
                }
            }
    } else if (valu_on) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
        }
    }
    const unsigned long long t1 = now();
    float s = acc0[0] + acc1[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wid] = t1 - t0;
}

static double median(std::vector<unsigned long long> v) {
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

int main() {
    int dev = 0;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, dev));
    const int cus = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.name, cus);
    unsigned long long* out;
    float* sink;
    float* dst;
    const size_t wg_stride = 1 << 20;
    CK(hipMalloc(&out, (size_t)cus * 32 * 8));
    CK(hipMalloc(&sink, 256));
    CK(hipMalloc(&dst, (size_t)cus * wg_stride));
    std::vector<unsigned long long> h((size_t)cus * 32);
    auto fetch = [&](int per_wg, int off, int nw) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, (size_t)cus * per_wg * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v;
        for (int b = 0; b < cus; ++b) {
            unsigned long long m = 0;
            for (int w = 0; w < nw; ++w) m = std::max(m, h[(size_t)b * per_wg + off + w]);
            v.push_back(m);
        }
        return median(v);
    };

    printf("== VALU: cycles per wave64 v_fma_f32 per SIMD (iters 2000 x 32 instr per wave)\n");
    for (int threads : {256, 512, 1024}) {
        const int wps = threads / 256;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_valu<8>, dim3(cus), dim3(threads), 0, 0, out, sink, 2000, 1.0001f);
            const double ci = fetch(16, 0, threads / 64);
            hipLaunchKernelGGL(k_valu<1>, dim3(cus), dim3(threads), 0, 0, out, sink, 2000, 1.0001f);
            const double cd = fetch(16, 0, threads / 64);
            if (rep) printf("  %d waves/SIMD: independent %.2f cyc per instr per SIMD (%.2f per wave), dependent chain %.2f (%.2f per wave)\n",
                            wps, ci / (2000.0 * 32 * wps), ci / (2000.0 * 32), cd / (2000.0 * 32 * wps), cd / (2000.0 * 32));
        }
    }

    printf("== stores: 512-thread workgroup per CU, per_wave instructions each; cycles per instruction (issue / acknowledged), B/clk per CU\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int nstore : {1, 2, 4, 8}) {
            const int per_wave = 40;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(cus), dim3(512), 0, 0, out, dst, nstore, per_wave, wg_stride);
                if (mode == 1) hipLaunchKernelGGL(k_store<1>, dim3(cus), dim3(512), 0, 0, out, dst, nstore, per_wave, wg_stride);
                if (mode == 2) hipLaunchKernelGGL(k_store<2>, dim3(cus), dim3(512), 0, 0, out, dst, nstore, per_wave, wg_stride);
                const double ti = fetch(32, 0, nstore), ta = fetch(32, 16, nstore);
                const double bytes = (double)nstore * per_wave * (mode == 1 ? 256 : 1024);
                if (rep) printf("  %s, %d waves storing: issue %.0f cyc/instr, acknowledged %.0f cyc/instr, %.1f B/clk/CU (all CUs at once)\n",
                                mode == 0 ? "b128 nt" : mode == 1 ? "b32" : "b128", nstore, ti / per_wave, ta / per_wave, bytes / ta);
            }
        }
    // the same with only a few CUs active (is it the CU or the memory side?)
    for (int grid : {8, 64}) {
        hipLaunchKernelGGL(k_store<0>, dim3(grid), dim3(512), 0, 0, out, dst, 8, 40, wg_stride);
        hipLaunchKernelGGL(k_store<0>, dim3(grid), dim3(512), 0, 0, out, dst, 8, 40, wg_stride);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), out, (size_t)grid * 32 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> v;
        for (int b = 0; b < grid; ++b) { unsigned long long m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, h[(size_t)b * 32 + 16 + w]); v.push_back(m); }
        printf("  b128 nt, 8 waves storing, only %d workgroups: %.1f B/clk/CU\n", grid, 8 * 40 * 1024.0 / median(v));
    }

    printf("== LDS reads: cycles per wave-instruction per CU (16 reads then lgkmcnt(0), 500 iters)\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int threads : {256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(cus), dim3(threads), 65536 + 16384, 0, out, sink, 500);
                if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(cus), dim3(threads), 65536 + 16384, 0, out, sink, 500);
                if (mode == 2) hipLaunchKernelGGL(k_lds<2>, dim3(cus), dim3(threads), 65536 + 16384, 0, out, sink, 500);
                const double c = fetch(16, 0, threads / 64);
                const double n = 500.0 * 16 * (threads / 64);
                if (rep) printf("  %s, %d waves/SIMD: %.2f cyc per instr per CU = %.0f B/clk\n",
                                mode == 0 ? "ds_read_b128" : mode == 1 ? "ds_read_b64_tr_b16" : "ds_read_b64", threads / 256,
                                c / n, (mode == 0 ? 1024.0 : 512.0) * n / c);
            }
        }

    printf("== MFMA (waves 0-3, 16 per iter) beside VALU (waves 4-7, 32 per iter), 2000 iters: cycles per iter\n");
    for (int cfg = 0; cfg < 3; ++cfg) {
        const int mf = cfg != 1, va = cfg != 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_mix, dim3(cus), dim3(512), 0, 0, out, sink, 2000, mf, va, 1.0001f);
            const double cm = fetch(16, 0, 4), cv = fetch(16, 4, 4);
            if (rep) printf("  mfma %d valu %d: MFMA waves %.1f cyc/iter (%.1f per MFMA), VALU waves %.1f cyc/iter (%.2f per instr)\n",
                            mf, va, cm / 2000, cm / 2000 / 16, cv / 2000, cv / 2000 / 32);
        }
    }
    return 0;
}
