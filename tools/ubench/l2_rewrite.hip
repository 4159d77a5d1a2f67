// Does a dirty line that is REWRITTEN in place stay in the XCD's L2 without fabric traffic?  (tools only.)
// The team kernel rewrites the same 0.52 MB of exchange lines per team every batch (18 us apart) and rocprofv3 shows all of
// them written to the fabric once per batch (WRITE_SIZE 1.81x the dE bytes).  Is that capacity -- E and dE streaming through
// the 4 MiB L2 in between -- or does the L2 clean dirty lines eagerly whatever the pressure?
// Each workgroup rewrites ITS OWN region of `kb` KB `iters` times, `gap` s_sleep units apart, and nothing else runs: with
// 256 workgroups x 64 KB the dirty set is 2 MiB per XCD, half an L2.  Run under
//   rocprofv3 --pmc WRITE_SIZE -- ./l2_rewrite          (and FETCH_SIZE in a second run)
// and compare WRITE_SIZE of each dispatch with (a) one copy of the set (write-back at the end only) and (b) iters copies.
//   hipcc -O3 --offload-arch=gfx950 -o l2_rewrite l2_rewrite.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// AUX: cache-policy bits of the stores (0 plain, 2 nt, 16 sc1).  STREAM_KB > 0: between two rewrites the workgroup also
// streams that many KB of nt loads from a buffer of its own far larger than the caches (the E stream of the real kernel).
template <int AUX>
__global__ __launch_bounds__(512) void k_rewrite(char* buf, int kb, int iters, int gap, const char* stream, size_t stream_bytes, int stream_kb, float* sink) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)blockIdx.x * kb * 1024, 0, kb * 1024, 0x00020000);
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
    float acc = 0.f;
    size_t spos = ((size_t)blockIdx.x * 7919u * 65536u) % stream_bytes;
    for (int it = 0; it < iters; ++it) {
        for (int off = threadIdx.x * 16; off < kb * 1024; off += 512 * 16) {
            x = x * 1664525u + 1013904223u;
            const u32x4 v = {x, x ^ 0x9E3779B9u, x * 2246822519u, (unsigned)it};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
        }
        if (stream_kb > 0) {
            const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream) + spos, 0, stream_kb * 1024, 0x00020000);
            for (int off = threadIdx.x * 16; off < stream_kb * 1024; off += 512 * 16) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r2, off, 0, 2);
                acc += __uint_as_float(v[0] & 0x3fffffffu);
            }
            spos = (spos + (size_t)gridDim.x * stream_kb * 1024) % (stream_bytes - (size_t)stream_kb * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(64);
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int kb = argc > 1 ? atoi(argv[1]) : 64, iters = argc > 2 ? atoi(argv[2]) : 64;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    char *buf, *stream;
    float* sink;
    const size_t sb = (size_t)4 << 30;
    CK(hipMalloc(&buf, (size_t)cus * kb * 1024));
    CK(hipMalloc(&stream, sb));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(stream, 1, sb));
    CK(hipDeviceSynchronize());
    printf("%d workgroups x %d KB rewritten %d times: one copy = %.1f MB, all copies = %.1f MB; dispatch order below:\n", cus, kb, iters,
           cus * kb / 1024.0, (double)cus * kb * iters / 1024.0);
    printf(" 1 plain gap 0 | 2 plain gap 40 (~18 us) | 3 nt gap 40 | 4 sc1 gap 40 | 5 plain gap 40 + 640 KB nt stream per rewrite | 6 the same, 64 KB stream\n");
    hipLaunchKernelGGL(k_rewrite<0>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 0, stream, sb, 0, sink);
    hipLaunchKernelGGL(k_rewrite<0>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 40, stream, sb, 0, sink);
    hipLaunchKernelGGL(k_rewrite<2>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 40, stream, sb, 0, sink);
    hipLaunchKernelGGL(k_rewrite<16>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 40, stream, sb, 0, sink);
    hipLaunchKernelGGL(k_rewrite<0>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 40, stream, sb, 640, sink);
    hipLaunchKernelGGL(k_rewrite<0>, dim3(cus), dim3(512), 0, 0, buf, kb, iters, 40, stream, sb, 64, sink);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
