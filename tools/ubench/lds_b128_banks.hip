// How does ds_read_b128 take its 64 lanes through the LDS banks?  One wave times a long run of ds_read_b128 for a table of
// per-lane address patterns (s_memtime around 4096 reads, 8 in flight) and prints cycles per read:
//   * the 32-row pattern of the tiled core's first form (known conflict-free from SQ_LDS_BANK_CONFLICT = 0),
//   * every LINEAR swizzle of the 16-row pattern of v_mfma_f32_16x16x32_f16 fragments -- lane (i = lane % 16, kg = lane / 16)
//     reads row i (64-byte rows), 16-byte piece kg ^ f(i), f(i) = (parity(i & a) << 1) | parity(i & b), a, b = 0..15,
//   * pairs: lane 0 and lane x swap their addresses inside the conflict-free 32-row pattern (no slowdown <=> same group).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench/lds_b128_banks tools/ubench/lds_b128_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_time(const unsigned* offs, int ncand, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[32768];
    const int lane = threadIdx.x;
    for (int i = lane; i < 32768 / 4; i += 64) reinterpret_cast<unsigned*>(sm)[i] = i;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)sm;
    for (int c = 0; c < ncand; ++c) {
        const unsigned a = base + offs[c * 64 + lane];
        uint4 acc = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < 512; ++it) {
            uint4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8\n\t"
                         "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7) : "v"(a) : "memory");
            acc.x += v0.x + v1.y + v2.z + v3.w + v4.x + v5.y + v6.z + v7.w;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[c] = t1 - t0;
        if (acc.x == 0x12345678u) out[ncand] = acc.x;
    }
}
static int par(int x) { return __builtin_popcount(x) & 1; }
int main() {
    std::vector<unsigned> offs;
    std::vector<const char*> kind;
    std::vector<int> pa, pb;
    auto push = [&](const char* k, int a, int b, auto f) {
        for (int l = 0; l < 64; ++l) offs.push_back((unsigned)f(l));
        kind.push_back(k); pa.push_back(a); pb.push_back(b);
    };
    auto old32 = [](int l) { const int row = l & 31; return row * 64 + 16 * ((l >> 5) ^ ((l >> 2) & 3)); };
    push("old 32-row pattern", 0, 0, old32);
    push("all lanes one address (broadcast)", 0, 0, [](int) { return 0; });
    push("lane-linear 16 B (1 KB contiguous)", 0, 0, [](int l) { return 16 * l; });
    for (int a = 0; a < 16; ++a)
        for (int b = 0; b < 16; ++b)
            push("16-row linear", a, b, [=](int l) { const int i = l & 15, kg = l >> 4; return i * 64 + 16 * (kg ^ ((par(i & a) << 1) | par(i & b))); });
    for (int x = 1; x < 64; ++x)
        push("old pattern, lanes 0 and x swapped", x, 0, [=](int l) { return old32(l == 0 ? x : l == x ? 0 : l); });
    const int n = (int)kind.size();
    unsigned* d_offs; unsigned long long* d_out;
    hipMalloc(&d_offs, offs.size() * 4); hipMalloc(&d_out, (n + 1) * 8);
    hipMemcpy(d_offs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
    k_time<<<1, 64>>>(d_offs, n, d_out);
    k_time<<<1, 64>>>(d_offs, n, d_out);
    std::vector<unsigned long long> out(n + 1);
    hipMemcpy(out.data(), d_out, (n + 1) * 8, hipMemcpyDeviceToHost);
    const double ref = (double)out[0];
    printf("s_memtime ticks for 4096 reads; ratio to the old 32-row pattern\n");
    for (int c = 0; c < 3; ++c) printf("%-40s %8llu  %.2f\n", kind[c], out[c], out[c] / ref);
    printf("16-row linear swizzles f(i) = (par(i&a) << 1) | par(i&b): ratio table, rows a = 0..15, columns b = 0..15\n");
    for (int a = 0; a < 16; ++a) {
        printf("a=%2d:", a);
        for (int b = 0; b < 16; ++b) printf(" %.2f", out[3 + a * 16 + b] / ref);
        printf("\n");
    }
    printf("old pattern with lanes 0 and x swapped (slot(x) != slot(0) unless marked =): ratio; ~1.00 <=> same group as lane 0\n");
    for (int x = 1; x < 64; ++x) {
        const bool same = ((old32(x) / 16) & 15) == ((old32(0) / 16) & 15);
        printf(" x=%2d%s %.2f%s", x, same ? "=" : " ", out[3 + 256 + x - 1] / ref, x % 8 == 7 ? "\n" : "");
    }
    printf("\n");
    return 0;
}
