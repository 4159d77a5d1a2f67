// Accuracy of the fused scale + split (ge2e_split_gemm.hpp: split4_scaled -- v_fma_mixlo / mixhi_f16): hi + lo against the exact
// product x * s in double, over 2^20 unit-row-like values and the kernels' 2^8 prescale (DESIGN.md hazard 26), next to the
// scale-then-split form it replaced.
// build: hipcc -O2 --offload-arch=gfx950 -I speaker_embedding_ge2e_loss_amd/csrc -o tools/ubench/mix_split_test tools/ubench/mix_split_test.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ge2e_common.hpp"
#include "ge2e_split_gemm.hpp"
using namespace ge2e;

__global__ void k_split(const float4* x, float sc, float4* fused, float4* two_step, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    h4 hi, lo;
    split4_scaled(x[i], sc, hi, lo);
    fused[i] = join4(hi, lo);
    const float4 s = make_float4(x[i].x * sc, x[i].y * sc, x[i].z * sc, x[i].w * sc);
    split4(s, hi, lo);
    two_step[i] = join4(hi, lo);
}
int main() {
    const int n = 1 << 18;
    std::vector<float4> x(n), a(n), b(n);
    srand(7);
    for (auto& v : x) {
        float t[4];
        for (float& e : t) e = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.25f * (rand() % 8 == 0 ? 1e-3f : 1.f);
        v = make_float4(t[0], t[1], t[2], t[3]);
    }
    float4 *dx, *da, *db;
    hipMalloc(&dx, n * 16); hipMalloc(&da, n * 16); hipMalloc(&db, n * 16);
    hipMemcpy(dx, x.data(), n * 16, hipMemcpyHostToDevice);
    const float sc = 256.f * 0.937f;     // prescale x a row's reciprocal norm
    k_split<<<n / 256, 256>>>(dx, sc, da, db, n);
    hipMemcpy(a.data(), da, n * 16, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 16, hipMemcpyDeviceToHost);
    // hi carries 11 bits; lo is an fp16 too: full relative accuracy (2^-22) where lo is a normal number, i.e. |x s| >= 1 (lo >=
    // 2^-14 ... ), and half a subnormal step (2^-25, in units of the SCALED value) absolute below that
    double ra = 0, rb = 0, aa = 0, ab = 0;
    for (int i = 0; i < n; ++i) {
        const float xs[4] = {x[i].x, x[i].y, x[i].z, x[i].w}, fa[4] = {a[i].x, a[i].y, a[i].z, a[i].w}, fb[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
        for (int k = 0; k < 4; ++k) {
            const double ex = (double)xs[k] * (double)sc;
            if (std::fabs(ex) >= 1.0) {
                ra = std::fmax(ra, std::fabs(fa[k] - ex) / std::fabs(ex));
                rb = std::fmax(rb, std::fabs(fb[k] - ex) / std::fabs(ex));
            } else {
                aa = std::fmax(aa, std::fabs(fa[k] - ex));
                ab = std::fmax(ab, std::fabs(fb[k] - ex));
            }
        }
    }
    printf("hi + lo against x * s, fused (v_fma_mix) | scale then split:\n  |x s| >= 1: worst relative error %.3e | %.3e (2^-22 = %.3e)\n"
           "  |x s| <  1: worst absolute error %.3e | %.3e (2^-25 = %.3e)\n", ra, rb, std::ldexp(1.0, -22), aa, ab, std::ldexp(1.0, -25));
    const double wa = ra;
    return wa < 2.5e-7 && aa < 6.2e-8 ? 0 : 1;
}
