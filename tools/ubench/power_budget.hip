// Socket power of ONE activity at a time on all CUs (tools only, not part of the product): what a wave-instruction, an MFMA,
// an LDS byte, an L2 byte and an HBM byte cost in joules on this MI355X, for the energy budget of the team kernels
// (DESIGN.md 3.2).  Each activity runs back to back for a few seconds while the host samples rocm-smi; the
// program prints rate, mean socket power over the steady part and (power - idle) / rate.
//   hipcc -O3 --offload-arch=gfx950 -o power_budget power_budget.hip && ./power_budget [seconds per activity]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- activities: every kernel is 512 threads (8 waves, 2 per SIMD), grid = CUs, `iters` trips ------------------------------
__global__ __launch_bounds__(512) void k_sleep(float* sink, int iters) {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
    if (sink == nullptr) return;
}
__global__ __launch_bounds__(512) void k_barrier(float* sink, int iters) {     // waves parked at s_barrier most of the time
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) for (int k = 0; k < 40; ++k) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
    }
    if (sink == nullptr) return;
}
// 32 independent-chain fp32 FMAs per trip
__global__ __launch_bounds__(512) void k_valu(float* sink, int iters, float a) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}
// the 4.5-cycle class: converts, packed ops, DPP
__global__ __launch_bounds__(512) void k_valu_cvt(float* sink, int iters, float a) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(512) void k_exp(float* sink, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}
// 16 MFMA 16x16x32 f16 per trip on WAVES waves of the workgroup (the others leave)
template <int WAVES>
__global__ __launch_bounds__(512) void k_mfma(float* sink, int iters) {
    if ((threadIdx.x >> 6) >= WAVES) return;
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}
// the same 16 MFMAs per trip on RANDOM operands that change from instruction to instruction (eight A and eight B fragments of
// hashed bits in [-1, 1), as unit-vector halves are): the loop above feeds ONE smooth pair again and again, and the matrix
// pipe's energy depends on how many operand bits toggle (MI355X_MICROARCH.md, DVFS give-back item 1: zeros ran +19 %)
__global__ __launch_bounds__(512) void k_mfma_rand(float* sink, int iters) {
    h8 a[8], b[8];
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            st = st * 1664525u + 1013904223u; a[k][i] = (_Float16)(((int)(st >> 8) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
            st = st * 1664525u + 1013904223u; b[k][i] = (_Float16)(((int)(st >> 8) & 0xFFFF) * (1.0f / 32768.0f) - 1.0f);
        }
    f32x4 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(4 * r + i) & 7], b[(4 * r + i + 3) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
}
// 16 ds_read_b128 per trip (conflict-free)
template <bool RANDOM>
__global__ __launch_bounds__(512) void k_lds(float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 512) {
        unsigned x = (unsigned)i * 2654435761u + blockIdx.x * 40503u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        reinterpret_cast<float*>(smem)[i] = RANDOM ? __uint_as_float((x & 0x007fffffu) | 0x3f000000u) : (float)i;   // random mantissas in [0.5, 1)
    }
    __syncthreads();
    const char* base = smem + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    f32x4 acc = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)(base) + (unsigned)((it & 3) * 8192)), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
// streaming 16-byte loads: every wave walks its own 1-KB rows of `buf` (span bytes, a power of two); AUX = cache policy.
// span >> L2 + Infinity Cache: HBM; span = 2 MB per XCD-ish: L2 hits.
template <int AUX>
__global__ __launch_bounds__(512) void k_read(const float* buf, size_t span_bytes, float* sink, int iters) {
    const size_t nrows = span_bytes / 1024;
    size_t row = ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 977u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, 0x7fffffff, 0x00020000);
    f32x4 acc = {};
    const unsigned lane_off = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const size_t r = (row + (size_t)i * 2048u) & (nrows - 1);
            if (span_bytes > (size_t)0x7fffffff) {      // 64-bit base per load: fold the row into a fresh resource
                const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf) + r * 256, 0, 1024, 0x00020000);
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r2, lane_off, 0, AUX);
            } else {
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(r * 1024) + lane_off, 0, AUX);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[0] += __uint_as_float(v[i][0]); acc[1] += __uint_as_float(v[i][3]); }
        row += 8u * 2048u + 1u;
    }
    if (acc[0] + acc[1] == 12345.678f) sink[0] = acc[0];
}

// sequential streams, the team kernels' pattern: workgroup b takes the 64-KB chunks b, b + grid, b + 2 grid, ... of the buffer
// (wave w of it the 8-KB piece w: eight 1-KB rows); MODE 0: nt loads, MODE 1: nt stores
template <int MODE>
__global__ __launch_bounds__(512) void k_stream(float* buf, size_t span_bytes, float* sink, int iters) {
    const size_t nchunks = span_bytes >> 16;
    f32x4 acc = {};
    const unsigned lane_off = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
    const u32x4 val = {1u, 2u, 3u, (unsigned)threadIdx.x};
    for (int it = 0; it < iters; ++it) {
        const size_t c = ((size_t)blockIdx.x + (size_t)it * gridDim.x) % nchunks;
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(buf) + (c << 16), 0, 65536, 0x00020000);
        if (MODE == 0) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r2, lane_off + 1024u * i, 0, 2);
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc[0] += __uint_as_float(v[i][0]); acc[1] += __uint_as_float(v[i][3]); }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_buffer_store_b128(val, r2, lane_off + 1024u * i, 0, 2);
        }
    }
    if (acc[0] + acc[1] == 12345.678f) sink[0] = acc[0];
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k_mfma32(float* sink, int iters) {
    h8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(512) void k_valu_mix(float* sink, int iters, float a) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_mix_f32 %0, %0, %1, %1 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(a));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}
__global__ __launch_bounds__(512) void k_valu_dpp(float* sink, int iters) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) sink[0] = s;
}

// ---- round 6: the activities the round-5 ledger left out (62-96 uJ of 465 per batch were unattributed) --------------------
// scalar ALU: 32 dependent-free s_add / s_mul / s_and / s_lshl per trip (the team kernel issues 435 scalar instructions per
// wave and batch: address arithmetic, buffer resources, loop control)
__global__ __launch_bounds__(512) void k_salu(float* sink, int iters, int a) {
    int x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(x0) : "s"(a) : "scc");
            asm volatile("s_mul_i32 %0, %0, %1" : "+s"(x1) : "s"(a) : "scc");
            asm volatile("s_and_b32 %0, %0, %1" : "+s"(x2) : "s"(a) : "scc");
            asm volatile("s_lshl_b32 %0, %0, 1" : "+s"(x3) : : "scc");
            asm volatile("s_add_u32 %0, %0, %1" : "+s"(x4) : "s"(a) : "scc");
            asm volatile("s_mul_i32 %0, %0, %1" : "+s"(x5) : "s"(a) : "scc");
            asm volatile("s_xor_b32 %0, %0, %1" : "+s"(x6) : "s"(a) : "scc");
            asm volatile("s_sub_u32 %0, %0, %1" : "+s"(x7) : "s"(a) : "scc");
        }
    }
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 123456789) sink[0] = 1.f;
}
// eight waves per CU standing at s_waitcnt behind ONE dependent L2 round trip each (a pointer chase through a 2-MiB span:
// every load's address comes from the previous load): what a wave costs while it waits for memory, and what the round
// trips themselves cost at this (low) rate
__global__ __launch_bounds__(512) void k_wait_l2(const unsigned* chain, float* sink, int iters) {
    unsigned idx = (blockIdx.x * 8u + (threadIdx.x >> 6)) * 4099u & 0x7FFFFu;        // 2 MiB of dwords
    for (int it = 0; it < iters; ++it) {
        unsigned v;
        asm volatile("global_load_dword %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(idx * 4u), "s"(chain) : "memory");
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)v) & 0x7FFFFu;
    }
    if (idx == 0x7FFFFFFFu) sink[0] = 1.f;
}
// the hand-off wait of the team kernels: ONE lane per workgroup polls a word in L2 (sc1 load, s_sleep 2 between polls) while
// the other seven waves stand at the workgroup barrier behind it
__global__ __launch_bounds__(512) void k_poll(const unsigned* word, float* sink, int polls) {
    if (threadIdx.x == 0) {
        unsigned seen = 0;
        for (int k = 0; k < polls; ++k) {
            seen += __hip_atomic_load(word + 32 * (blockIdx.x & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(2);
        }
        if (seen == 0x7FFFFFFFu) sink[0] = 1.f;
    }
    __syncthreads();
}
// workgroup barriers back to back: all eight waves arrive, leave, arrive ... (the training kernel has ten per batch)
__global__ __launch_bounds__(512) void k_barrier_rate(float* sink, int iters) {
    for (int it = 0; it < iters; ++it) __syncthreads();
    if (sink == nullptr) return;
}
// the streams above read what hipMemset left (zeros) and store four small constants: real embeddings toggle every data line.
// k_fill writes hashed bits so that the read activities can be repeated on random data
__global__ __launch_bounds__(512) void k_fill(unsigned* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) {
        unsigned x = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        buf[i] = x;
    }
}
__global__ __launch_bounds__(512) void k_stream_store_rand(float* buf, size_t span_bytes, int iters) {
    const size_t nchunks = span_bytes >> 16;
    const unsigned lane_off = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 7u;
    for (int it = 0; it < iters; ++it) {
        const size_t c = ((size_t)blockIdx.x + (size_t)it * gridDim.x) % nchunks;
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(buf) + (c << 16), 0, 65536, 0x00020000);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            const u32x4 val = {x, x * 2246822519u, x ^ 0x9E3779B9u, x * 3266489917u};
            __builtin_amdgcn_raw_buffer_store_b128(val, r2, lane_off + 1024u * i, 0, 2);
        }
    }
}
// ---- host: run an activity for `secs`, sample rocm-smi meanwhile -----------------------------------------------------------
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

struct Result { double secs, launches, watts, mhz; };
template <class F>
static Result run(double secs, F&& launch) {
    std::atomic<bool> stop{false};
    std::vector<double> ws, fs;
    std::thread sampler([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(700));      // let the power settle
        while (!stop.load()) {
            double w, m;
            if (smi(w, m)) { ws.push_back(w); fs.push_back(m); }
            std::this_thread::sleep_for(std::chrono::milliseconds(150));
        }
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
    return {el, n, ws.empty() ? 0 : w / ws.size(), fs.empty() ? 0 : m / fs.size()};
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float* sink;
    CK(hipMalloc(&sink, 256));
    const size_t big = (size_t)8 << 30;          // 8 GiB: HBM stream
    float* buf;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 0, big));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const double waves = (double)cus * 8;
    printf("MI355X, %d CUs, %.1f s per activity; socket power from rocm-smi (mean of the steady samples)\n", cus, secs);
    std::this_thread::sleep_for(std::chrono::seconds(1));
    double idle_w = 0, idle_m = 0;
    { double w = 0, m = 0; int n = 0; for (int i = 0; i < 6; ++i) { double a, b; if (smi(a, b)) { w += a; m += b; ++n; } std::this_thread::sleep_for(std::chrono::milliseconds(200)); } idle_w = w / (n ? n : 1); idle_m = m / (n ? n : 1); }
    printf("%-44s %8.0f W  %5.0f MHz\n", "idle (no kernel)", idle_w, idle_m);
    auto report = [&](const char* name, const Result& r, double units_per_launch, const char* unit, double scale, const char* eunit) {
        const double rate = units_per_launch * r.launches / r.secs;
        printf("%-44s %8.0f W  %5.0f MHz  %10.3e %s/s  -> %8.2f %s above idle\n", name, r.watts, r.mhz, rate, unit,
               (r.watts - idle_w) / rate * scale, eunit);
        fflush(stdout);
    };
    const int IT = 20000;
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_sleep, dim3(cus), dim3(512), 0, 0, sink, 4000); });
      report("8 waves per CU in s_sleep", r, waves, "wave-launches", 1.0, "J per wave-launch"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_barrier, dim3(cus), dim3(512), 0, 0, sink, 100); });
      report("7 of 8 waves per CU parked at s_barrier", r, waves, "wave-launches", 1.0, "J per wave-launch"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_valu, dim3(cus), dim3(512), 0, 0, sink, IT, 1.0001f); });
      report("v_fma_f32, 2 waves per SIMD", r, waves * IT * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_salu, dim3(cus), dim3(512), 0, 0, sink, IT, 3); });
      report("scalar ALU (s_add / s_mul / s_and ...), 8 waves per CU", r, waves * IT * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_barrier_rate, dim3(cus), dim3(512), 0, 0, sink, IT * 4); });
      report("s_barrier back to back, 8 waves per CU", r, (double)cus * IT * 4.0, "barriers", 1e9, "nJ per workgroup barrier"); }
    { unsigned* chain; CK(hipMalloc(&chain, 1u << 21));
      std::vector<unsigned> hc(1u << 19); for (unsigned i = 0; i < hc.size(); ++i) hc[i] = (i * 2654435761u + 12345u) & 0x7FFFFu;
      CK(hipMemcpy(chain, hc.data(), 1u << 21, hipMemcpyHostToDevice));
      { auto r = run(secs, [&] { hipLaunchKernelGGL(k_wait_l2, dim3(cus), dim3(512), 0, 0, chain, sink, 4000); });
        report("8 waves per CU at s_waitcnt, one L2 round trip each", r, waves * 4000.0, "round trips", 1e9, "nJ per wave round trip"); }
      { auto r = run(secs, [&] { hipLaunchKernelGGL(k_poll, dim3(cus), dim3(512), 0, 0, chain, sink, 4000); });
        report("one lane per CU polls L2 (sc1 + s_sleep 2), 7 waves at the barrier", r, (double)cus * 4000.0, "polls", 1e9, "nJ per poll (incl. the waiting waves)"); }
      CK(hipFree(chain)); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_valu_cvt, dim3(cus), dim3(512), 0, 0, sink, IT, 1.0001f); });
      report("v_cvt_pk_f16_f32, 2 waves per SIMD", r, waves * IT * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_valu_mix, dim3(cus), dim3(512), 0, 0, sink, IT, 1.0001f); });
      report("v_fma_mix_f32, 2 waves per SIMD", r, waves * IT * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_valu_dpp, dim3(cus), dim3(512), 0, 0, sink, IT); });
      report("v_add_f32_dpp, 2 waves per SIMD", r, waves * IT * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_exp, dim3(cus), dim3(512), 0, 0, sink, IT / 2); });
      report("v_exp_f32, 2 waves per SIMD", r, waves * (IT / 2) * 32.0, "wave-instr", 1e9, "nJ per wave-instr"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_mfma<4>, dim3(cus), dim3(512), 0, 0, sink, IT / 2); });
      report("mfma 16x16x32 f16, 1 wave per SIMD", r, cus * 4.0 * (IT / 2) * 16.0, "MFMA", 1e9, "nJ per MFMA (16.4 kFLOP)"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_mfma<8>, dim3(cus), dim3(512), 0, 0, sink, IT / 4); });
      report("mfma 16x16x32 f16, 2 waves per SIMD", r, cus * 8.0 * (IT / 4) * 16.0, "MFMA", 1e9, "nJ per MFMA (16.4 kFLOP)"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_mfma_rand, dim3(cus), dim3(512), 0, 0, sink, IT / 4); });
      report("mfma 16x16x32 f16, RANDOM operands, 2 waves/SIMD", r, cus * 8.0 * (IT / 4) * 16.0, "MFMA", 1e9, "nJ per MFMA (16.4 kFLOP)"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_mfma32, dim3(cus), dim3(512), 0, 0, sink, IT / 8); });
      report("mfma 32x32x16 f16, 2 waves per SIMD", r, cus * 8.0 * (IT / 8) * 8.0, "MFMA", 1e9, "nJ per MFMA (32.8 kFLOP)"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_lds<false>, dim3(cus), dim3(512), 65536, 0, sink, IT / 2); });
      report("ds_read_b128, 2 waves per SIMD", r, waves * (IT / 2) * 16.0 * 1024.0, "B", 1e12, "pJ per LDS byte"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_lds<true>, dim3(cus), dim3(512), 65536, 0, sink, IT / 2); });
      report("ds_read_b128, RANDOM data, 2 waves per SIMD", r, waves * (IT / 2) * 16.0 * 1024.0, "B", 1e12, "pJ per LDS byte"); }
    { const size_t span = (size_t)1 << 21;       // 2 MiB: stays in every XCD's L2
      auto r = run(secs, [&] { hipLaunchKernelGGL(k_read<0>, dim3(cus), dim3(512), 0, 0, buf, span, sink, 400); });
      report("16-B loads, 2 MiB span (L2 hits)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { const size_t span = (size_t)1 << 27;       // 128 MiB: Infinity Cache
      auto r = run(secs, [&] { hipLaunchKernelGGL(k_read<0>, dim3(cus), dim3(512), 0, 0, buf, span, sink, 400); });
      report("16-B loads, 128 MiB span (Infinity Cache)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_read<2>, dim3(cus), dim3(512), 0, 0, buf, big, sink, 400); });
      report("16-B nt loads, 8 GiB span (HBM)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_stream<0>, dim3(cus), dim3(512), 0, 0, buf, big, sink, 400); });
      report("sequential 64-KB chunks, nt loads (HBM)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_stream<1>, dim3(cus), dim3(512), 0, 0, buf, big, sink, 400); });
      report("sequential 64-KB chunks, nt stores (HBM)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    // the HBM streams again on RANDOM data (the passes above moved zeros / constants)
    hipLaunchKernelGGL(k_fill, dim3(cus * 8), dim3(512), 0, 0, reinterpret_cast<unsigned*>(buf), big / 4);
    CK(hipDeviceSynchronize());
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_stream<0>, dim3(cus), dim3(512), 0, 0, buf, big, sink, 400); });
      report("RANDOM data: sequential 64-KB chunks, nt loads (HBM)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { auto r = run(secs, [&] { hipLaunchKernelGGL(k_stream_store_rand, dim3(cus), dim3(512), 0, 0, buf, big, 400); });
      report("RANDOM data: sequential 64-KB chunks, nt stores (HBM)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    { const size_t span = (size_t)1 << 21;
      auto r = run(secs, [&] { hipLaunchKernelGGL(k_read<0>, dim3(cus), dim3(512), 0, 0, buf, span, sink, 400); });
      report("RANDOM data: 16-B loads, 2 MiB span (L2 hits)", r, waves * 400.0 * 8.0 * 1024.0, "B", 1e12, "pJ per byte"); }
    // "hot idle": the resident-waves point again, right behind ten seconds of matrix work at the power limit -- leakage at the
    // loaded die temperature is part of what a kernel at the cap pays per second, and the first point above was taken cold
    { run(10.0, [&] { hipLaunchKernelGGL(k_mfma<8>, dim3(cus), dim3(512), 0, 0, sink, IT / 4); });
      auto r = run(2.0, [&] { hipLaunchKernelGGL(k_sleep, dim3(cus), dim3(512), 0, 0, sink, 4000); });
      report("HOT: 8 waves per CU in s_sleep, behind 10 s of MFMA", r, waves, "wave-launches", 1.0, "J per wave-launch");
      auto r2 = run(2.0, [&] { hipLaunchKernelGGL(k_barrier, dim3(cus), dim3(512), 0, 0, sink, 100); });
      report("HOT: 7 of 8 waves parked at s_barrier (2 s later)", r2, waves, "wave-launches", 1.0, "J per wave-launch"); }
    return 0;
}
