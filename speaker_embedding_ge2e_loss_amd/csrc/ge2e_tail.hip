// The callers either side of the loss: the encoder's tail on the input side and the equal-error-rate
// threshold sweep on the output side.  Both are HBM/L2-bound row work, one wave per row; no MFMA.
#include "ge2e_common.hpp"
#include "ge2e_tail.hpp"

namespace ge2e {

namespace {

// ---- encoder tail: L2-normalise (s2:34) + un-permute (s4:186) + (N,M,D) layout (s4:189) in one pass ---------------
// out row i = y[src[i]] / |y[src[i]]|  (src = the reference's `unperm` list).  rn[i] = 1 / |y[src[i]]| is kept for the
// backward.  Like the reference there is no epsilon: a zero row divides by zero (inf/nan), s2:34.
__global__ __launch_bounds__(256) void tail_fwd_kernel(const float* __restrict__ y, const int* __restrict__ src, int rows,
                                                       int D, int vec, float* __restrict__ e, float* __restrict__ rn) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, lane = threadIdx.x % kWave;
    if (wave >= rows) return;
    const int s = src ? src[wave] : wave;
    if ((unsigned)s >= (unsigned)rows) return;  // not a permutation: the host validated, this only keeps the access in range
    const float* yr = y + (size_t)s * D;
    float* er = e + (size_t)wave * D;
    if (vec) {  // row held in registers: one read of y
        float4 v[4];
        float sq = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (c * kWave + lane) * 4;
            v[c] = d < D ? *reinterpret_cast<const float4*>(yr + d) : make_float4(0.f, 0.f, 0.f, 0.f);
            sq += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
        }
        sq = wave_sum(sq);
        const float n = sqrtf(sq);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int d = (c * kWave + lane) * 4;
            if (d < D) *reinterpret_cast<float4*>(er + d) = make_float4(v[c].x / n, v[c].y / n, v[c].z / n, v[c].w / n);
        }
        if (lane == 0) rn[wave] = 1.0f / n;
        return;
    }
    float sq = 0.f;
    for (int d = lane; d < D; d += kWave) sq += yr[d] * yr[d];
    sq = wave_sum(sq);
    const float n = sqrtf(sq);
    for (int d = lane; d < D; d += kWave) er[d] = yr[d] / n;
    if (lane == 0) rn[wave] = 1.0f / n;
}

// backward: dy[src[i]] = (g_i - e_i (e_i . g_i)) * rn_i.  src is a permutation, so every row of dy is written once.
__global__ __launch_bounds__(256) void tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ e,
                                                       const float* __restrict__ rn, const int* __restrict__ src, int rows,
                                                       int D, float* __restrict__ dy) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave, lane = threadIdx.x % kWave;
    if (wave >= rows) return;
    const int s = src ? src[wave] : wave;
    if ((unsigned)s >= (unsigned)rows) return;
    const float* gr = g + (size_t)wave * D;
    const float* er = e + (size_t)wave * D;
    float* dr = dy + (size_t)s * D;
    float dot = 0.f;
    for (int d = lane; d < D; d += kWave) dot += gr[d] * er[d];
    dot = wave_sum(dot);
    const float r = rn[wave];
    for (int d = lane; d < D; d += kWave) dr[d] = (gr[d] - er[d] * dot) * r;
}

// ---- equal-error-rate sweep (s5:57-98) -----------------------------------------------------------------------------
// For every threshold t: fa[t] = #{(j,i,k), k != j : S[j][i][k] > thr[t]} (s5:82: sum(S_thres[i]) - sum(S_thres[i,:,i]))
// and ta[t] = #{(j,i) : S[j][i][j] > thr[t]} (s5:89 counts M - that).  thr is non-decreasing, so each entry is binned
// once by the number of thresholds it exceeds (binary search on the fp32 table: the comparison itself is the reference's
// fp32 `S > thres`), LDS histograms, then a suffix sum.  One workgroup per batch.  counts [B][T][2] int32.
__global__ __launch_bounds__(256) void eer_counts_kernel(const float* __restrict__ S, int N, int M,
                                                         const float* __restrict__ thr, int T, int* __restrict__ counts) {
    extern __shared__ int hist[];  // [2][T + 1]
    int* hfa = hist;
    int* hta = hist + (T + 1);
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < 2 * (T + 1); t += blockDim.x) hist[t] = 0;
    __syncthreads();
    const size_t total = (size_t)N * M * N;
    const float* s = S + (size_t)b * total;
    for (size_t idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const float v = s[idx];
        const int k = (int)(idx % N), j = (int)(idx / ((size_t)M * N));
        int lo = 0, hi = T;  // number of thresholds below v: first t with !(v > thr[t])
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (v > thr[mid]) lo = mid + 1; else hi = mid;
        }
        atomicAdd(k == j ? &hta[lo] : &hfa[lo], 1);
    }
    __syncthreads();
    // counts[t] = number of entries whose bin is > t
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        int fa = 0, ta = 0;
        for (int u = t + 1; u <= T; ++u) { fa += hfa[u]; ta += hta[u]; }
        counts[((size_t)b * T + t) * 2] = fa;
        counts[((size_t)b * T + t) * 2 + 1] = ta;
    }
}

// ---- batch sampler (s1_dataset_loader.py:65-77) ---------------------------------------------------------------------
// The reference keeps one (U, T, F) float64 array per speaker on disk, draws M utterance indices (with replacement) and
// one crop start per speaker on the host, slices, stacks, and casts to float32 at the encoder's door (s2:28).  Here the
// arrays stay resident in HBM in their on-disk type and one launch gathers + casts the (N, M, L, F) batch:
//   out[n][m][l][f] = (float) store[spk_off[n] + (utt[n][m] * T + clip[n] + l) * F + f]
// One thread per output element, consecutive threads along (l, f): the L * F crop of an utterance is contiguous in the
// store, so reads and writes are both fully coalesced.  HBM-bound: 8 (or 4) bytes in, 4 out per element.
template <typename T>
__global__ __launch_bounds__(256) void sample_batch_kernel(const T* __restrict__ store, const long long* __restrict__ spk_off,
                                                           const int* __restrict__ utt, const int* __restrict__ clip,
                                                           int M, int Tfr, int L, int F, float* __restrict__ out) {
    const int nm = blockIdx.y, n = nm / M;
    const size_t crop = (size_t)L * F;
    const T* src = store + spk_off[n] + ((size_t)utt[nm] * Tfr + clip[n]) * F;
    float* dst = out + (size_t)nm * crop;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < crop; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (float)src[i];
}

}  // namespace

hipError_t launch_sample_batch(const void* store, int is_f64, const long long* spk_off, const int* utt, const int* clip,
                               int N, int M, int Tfr, int L, int F, float* out, hipStream_t stream) {
    const size_t crop = (size_t)L * F;
    unsigned gx = (unsigned)((crop + 255) / 256);
    if (gx > 64) gx = 64;
    const dim3 grid(gx, (unsigned)(N * M));
    if (is_f64)
        sample_batch_kernel<double><<<grid, 256, 0, stream>>>(static_cast<const double*>(store), spk_off, utt, clip, M, Tfr, L, F, out);
    else
        sample_batch_kernel<float><<<grid, 256, 0, stream>>>(static_cast<const float*>(store), spk_off, utt, clip, M, Tfr, L, F, out);
    return hipGetLastError();
}

hipError_t launch_tail_fwd(const float* y, const int* src, int rows, int D, float* e, float* rn, hipStream_t stream) {
    const int wpb = 256 / kWave;
    const int vec = (D & 3) == 0 && D <= 4 * kWave * 4 && (((uintptr_t)y | (uintptr_t)e) & 15) == 0;
    tail_fwd_kernel<<<(rows + wpb - 1) / wpb, 256, 0, stream>>>(y, src, rows, D, vec, e, rn);
    return hipGetLastError();
}

hipError_t launch_tail_bwd(const float* g, const float* e, const float* rn, const int* src, int rows, int D, float* dy,
                           hipStream_t stream) {
    const int wpb = 256 / kWave;
    tail_bwd_kernel<<<(rows + wpb - 1) / wpb, 256, 0, stream>>>(g, e, rn, src, rows, D, dy);
    return hipGetLastError();
}

hipError_t launch_eer_counts(const float* S, int B, int N, int M, const float* thr, int T, int* counts,
                             hipStream_t stream) {
    eer_counts_kernel<<<B, 256, 2 * (size_t)(T + 1) * sizeof(int), stream>>>(S, N, M, thr, T, counts);
    return hipGetLastError();
}

}  // namespace ge2e
