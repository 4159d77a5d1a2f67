// Launchers of the kernels either side of the loss (ge2e_tail.hip): encoder tail, equal-error-rate sweep.
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

hipError_t launch_tail_fwd(const float* y, const int* src, int rows, int D, float* e, float* rn, hipStream_t stream);
hipError_t launch_tail_bwd(const float* g, const float* e, const float* rn, const int* src, int rows, int D, float* dy,
                           hipStream_t stream);
hipError_t launch_eer_counts(const float* S, int B, int N, int M, const float* thr, int T, int* counts,
                             hipStream_t stream);

hipError_t launch_sample_batch(const void* store, int is_f64, const long long* spk_off, const int* utt, const int* clip,
                               int N, int M, int Tfr, int L, int F, float* out, hipStream_t stream);

}  // namespace ge2e
