// Which f16 MFMA shape should the tiled contractions use?  The same 128 x 64 wave tile (2 x 4 waves = a 256 x 256 workgroup
// tile, K-step 32, hi/lo split = 3 MFMAs per product, operand fragments from LDS by ds_read_b128) on
//   v_mfma_f32_32x32x16_f16  (4 x 2 blocks per wave, 2 K-slices per step:  48 MFMAs of 32 K FLOP ... per K-step 12 fragment reads)
//   v_mfma_f32_16x16x32_f16  (8 x 4 blocks per wave, 1 K-slice  per step:  96 MFMAs of 16 K FLOP ... per K-step 24 fragment reads)
// -- same FLOPs, same LDS bytes, same accumulator registers.  Prints TFLOP/s, clock and socket power of each (rocm-smi sampled
// while the loop runs): the 32 x 32 shape is clocked down much harder by the power management (tools/ubench/power_budget.hip
// shows it for bare MFMA loops; this is the version with the fragment traffic of the real K loop).
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shapes mfma_shapes.hip && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS stage: A [256 rows][32 k] hi, lo; B [256 cols][32 k] hi, lo (pitch 40 halfs: conflict-free b128 rows) = 80 KB
constexpr int KP = 40;
__device__ __forceinline__ h8 frag(const _Float16* p) { return *reinterpret_cast<const h8*>(p); }

template <int SHAPE>
__global__ __launch_bounds__(512) void k_tile(float* sink, int ksteps) {
    extern __shared__ __attribute__((aligned(16))) _Float16 sm[];
    for (int i = threadIdx.x; i < 4 * 256 * KP; i += 512) sm[i] = (_Float16)(0.001f * (i & 255));
    __syncthreads();
    const _Float16 *Ah = sm, *Al = sm + 256 * KP, *Bh = sm + 2 * 256 * KP, *Bl = sm + 3 * 256 * KP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wa = wid >> 2, wb = wid & 3;
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4][2] = {};
        const int ra = 128 * wa + (lane & 31), rb = 64 * wb + (lane & 31), ko = 8 * (lane >> 5);
        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                h8 bh[2], bl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) { bh[u] = frag(Bh + (rb + 32 * u) * KP + 16 * sl + ko); bl[u] = frag(Bl + (rb + 32 * u) * KP + 16 * sl + ko); }
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const h8 ah = frag(Ah + (ra + 32 * a) * KP + 16 * sl + ko), al = frag(Al + (ra + 32 * a) * KP + 16 * sl + ko);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        acc[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[u], acc[a][u], 0, 0, 0);
                        acc[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[u], acc[a][u], 0, 0, 0);
                        acc[a][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[u], acc[a][u], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) s += acc[a][u][i];
    } else {
        f32x4 acc[8][4] = {};
        const int ra = 128 * wa + (lane & 15), rb = 64 * wb + (lane & 15), ko = 8 * (lane >> 4);
        for (int ks = 0; ks < ksteps; ++ks) {
            h8 bh[4], bl[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { bh[u] = frag(Bh + (rb + 16 * u) * KP + ko); bl[u] = frag(Bl + (rb + 16 * u) * KP + ko); }
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const h8 ah = frag(Ah + (ra + 16 * a) * KP + ko), al = frag(Al + (ra + 16 * a) * KP + ko);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[a][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u], acc[a][u], 0, 0, 0);
                    acc[a][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[u], acc[a][u], 0, 0, 0);
                    acc[a][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[u], acc[a][u], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int u = 0; u < 4; ++u) s += acc[a][u][0] + acc[a][u][1] + acc[a][u][2] + acc[a][u][3];
    }
    if (s == 12345.678f) sink[0] = s;
}

static bool smi(double& watts, double& mhz) {
    FILE* f = popen("/opt/rocm/bin/rocm-smi --showclocks --showpower -d 0 2>/dev/null", "r");
    if (!f) return false;
    char line[512];
    watts = mhz = 0;
    while (fgets(line, sizeof line, f)) {
        if (strstr(line, "Power (W)")) { const char* c = strrchr(line, ':'); if (c) watts = atof(c + 1); }
        if (strstr(line, "sclk")) { const char* c = strchr(line, '('); if (c) mhz = atof(c + 1); }
    }
    pclose(f);
    return watts > 0;
}
template <class F>
static void run(const char* name, double secs, double flop_per_launch, F&& launch) {
    std::atomic<bool> stop{false};
    std::vector<double> ws, fs;
    std::thread sampler([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(700));
        while (!stop.load()) { double w, m; if (smi(w, m)) { ws.push_back(w); fs.push_back(m); } std::this_thread::sleep_for(std::chrono::milliseconds(150)); }
    });
    const auto t0 = std::chrono::steady_clock::now();
    double n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int k = 0; k < 8; ++k) launch();
        CK(hipDeviceSynchronize());
        n += 8;
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    sampler.join();
    double w = 0, m = 0;
    for (double x : ws) w += x;
    for (double x : fs) m += x;
    printf("%-60s %8.1f TFLOP/s (f16 MFMA issued)  %6.0f MHz  %6.0f W\n", name, flop_per_launch * n / el / 1e12, fs.empty() ? 0 : m / fs.size(), ws.empty() ? 0 : w / ws.size());
    fflush(stdout);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* sink;
    CK(hipMalloc(&sink, 256));
    const int lds = 4 * 256 * KP * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<16>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int ksteps = 4000;
    const double flop = (double)cus * 256.0 * 256.0 * 32.0 * 2.0 * 3.0 * ksteps;      // per launch, issued (3 products)
    printf("256 x 256 x 32 split-fp16 K-step from LDS, one workgroup of 8 waves per CU, %d CUs\n", cus);
    run("v_mfma_f32_32x32x16_f16 (the tiled core's shape)", secs, flop, [&] { hipLaunchKernelGGL(k_tile<32>, dim3(cus), dim3(512), lds, 0, sink, ksteps); });
    run("v_mfma_f32_16x16x32_f16", secs, flop, [&] { hipLaunchKernelGGL(k_tile<16>, dim3(cus), dim3(512), lds, 0, sink, ksteps); });
    run("v_mfma_f32_32x32x16_f16 (again)", secs, flop, [&] { hipLaunchKernelGGL(k_tile<32>, dim3(cus), dim3(512), lds, 0, sink, ksteps); });
    run("v_mfma_f32_16x16x32_f16 (again)", secs, flop, [&] { hipLaunchKernelGGL(k_tile<16>, dim3(cus), dim3(512), lds, 0, sink, ksteps); });
    return 0;
}
