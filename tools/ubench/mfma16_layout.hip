// Layout checks behind ge2e_tiled.hip's LDS-DMA core: (1) the operand / result layout of v_mfma_f32_16x16x32_f16 and
// (2) the two-swap conversion of four 16 x 16 accumulator blocks into one 32 x 32 accumulator block.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/mfma16_layout tools/ubench/mfma16_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const _Float16* A, const _Float16* B, float* D) {   // A [16][32], B [32][16] row-major, D [16][16]
    const int l = threadIdx.x, i = l & 15, kg = l >> 4;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 32 + 8 * kg + e]; b[e] = B[(8 * kg + e) * 16 + i]; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * kg + j) * 16 + i] = c[j];
}
__global__ void k_conv(float* Y) {   // Y [64 lanes][16 regs]
    const int l = threadIdx.x;
    f4 X[2][2];
    for (int ar = 0; ar < 2; ++ar)
        for (int bc = 0; bc < 2; ++bc)
            for (int j = 0; j < 4; ++j) X[ar][bc][j] = (float)((16 * ar + 4 * (l >> 4) + j) * 100 + 16 * bc + (l & 15));
    float acc[16];
    for (int ar = 0; ar < 2; ++ar)
        for (int j = 0; j < 4; ++j) {
            const float pv = X[ar][0][j], qv = X[ar][1][j];
            const auto s1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(pv), __float_as_uint(qv), false, false);
            const auto s2 = __builtin_amdgcn_permlane32_swap(s1[0], s1[1], false, false);
            acc[4 * (2 * ar) + j] = __uint_as_float(s2[0]);
            acc[4 * (2 * ar + 1) + j] = __uint_as_float(s2[1]);
        }
    for (int r = 0; r < 16; ++r) Y[l * 16 + r] = acc[r];
}
int main() {
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    for (auto& x : A) x = (_Float16)((rand() % 17) - 8);
    for (auto& x : B) x = (_Float16)((rand() % 17) - 8);
    _Float16 *dA, *dB; float *dD, *dY;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, 256 * 4); hipMalloc(&dY, 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD);
    k_conv<<<1, 64>>>(dY);
    std::vector<float> D(256), Y(1024);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float s = 0;
            for (int k = 0; k < 32; ++k) s += (float)A[i * 32 + k] * (float)B[k * 16 + j];
            if (s != D[i * 16 + j]) ++bad;
        }
    printf("mfma 16x16x32 layout: %d of 256 wrong\n", bad);
    int badc = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int g = r / 4, j = r % 4, h = l / 32, c = l % 32;
            const float want = (float)((8 * g + 4 * h + j) * 100 + c);
            if (Y[l * 16 + r] != want) { if (badc < 8) printf("  lane %d reg %d: got %g want %g\n", l, r, Y[l * 16 + r], want); ++badc; }
        }
    printf("16x16 -> 32x32 accumulator conversion: %d of 1024 wrong\n", badc);
    return bad || badc;
}
