// What does it cost a workgroup to re-read ITS OWN operand block once per row tile, when every CU does so?  (tools only.)
// The one-workgroup-per-batch kernel for N = 256 cannot hold its batch's unit centroids on chip: 256 x 256 x
// (hi + lo) fp16 = 256 KB against 160 KB of LDS -- so each workgroup would stream its own 256 KB again for every row tile, 32
// different blocks per XCD (8 MB against a 4 MiB L2), next to the tile's own rows from HBM.  This loop is that traffic and
// nothing else: per trip the workgroup reads its `kb` KB block (default cache policy, 16-byte loads, eight in flight per lane)
// and `tile_kb` KB of a stream that is never re-read (nt).  Printed: microseconds per trip and GB/s per CU; under
//   rocprofv3 --pmc FETCH_SIZE -- ./chat_restream          (x2: MI355X_MICROARCH.md, HBM)
// the fabric bytes per dispatch say how much of the re-read the L2 kept.
//   hipcc -O3 --offload-arch=gfx950 -o chat_restream chat_restream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_restream(const char* blocks, int kb, const char* stream, size_t stream_bytes, int tile_kb,
                                                  int iters, float* sink) {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blocks) + (size_t)blockIdx.x * kb * 1024, 0, kb * 1024, 0x00020000);
    float acc = 0.f;
    size_t spos = ((size_t)blockIdx.x * 7919u * 131072u) % (stream_bytes / 2);
    const unsigned lane_off = threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
        for (int base = 0; base < kb * 1024; base += 8 * 8192) {          // 64 KB per pass of the workgroup, 8 loads per lane
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, base + i * 8192 + lane_off, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += __uint_as_float(v[i][0] & 0x3fffffffu);
        }
        if (tile_kb > 0) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream) + spos, 0, tile_kb * 1024, 0x00020000);
            for (int base = 0; base < tile_kb * 1024; base += 8 * 8192) {
                u32x4 v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + i * 8192 + lane_off, 0, 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += __uint_as_float(v[i][0] & 0x3fffffffu);
            }
            spos = (spos + (size_t)gridDim.x * tile_kb * 1024) % (stream_bytes - (size_t)tile_kb * 1024);
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    char *blocks, *stream;
    float* sink;
    const size_t sb = (size_t)8 << 30;
    CK(hipMalloc(&blocks, (size_t)cus * 512 * 1024));
    CK(hipMalloc(&stream, sb));
    CK(hipMalloc(&sink, 256));
    CK(hipMemset(blocks, 1, (size_t)cus * 512 * 1024));
    CK(hipMemset(stream, 2, sb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%d workgroups (one per CU), %d trips each; per trip: the workgroup's own block + a fresh nt tile\n", cus, iters);
    const int cases[][2] = {{256, 0}, {256, 128}, {512, 128}, {128, 128}, {64, 128}, {0, 128}, {256, 64}};
    for (auto& c : cases) {
        const int kb = c[0], tile = c[1];
        hipLaunchKernelGGL(k_restream, dim3(cus), dim3(512), 0, 0, blocks, kb, stream, sb, tile, 20, sink);   // warm-up
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_restream, dim3(cus), dim3(512), 0, 0, blocks, kb, stream, sb, tile, iters, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, bytes = (kb + tile) * 1024.0;
        printf("block %3d KB (%5.1f MB per XCD) + tile %3d KB: %7.2f us per trip = %6.1f GB/s per CU = %5.2f TB/s chip-wide\n", kb,
               kb * 32 / 1024.0, tile, us, bytes / us * 1e-3, bytes * cus / us * 1e-6);
    }
    return 0;
}
