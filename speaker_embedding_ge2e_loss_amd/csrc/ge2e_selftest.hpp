#pragma once
#include "ge2e_common.hpp"
namespace ge2e {
hipError_t launch_selftest_split(const float* A, const float* Bm, const float* G, float* X, float* GE, float* GC,
                                 hipStream_t stream);
hipError_t launch_selftest_wave(const float* x, float* out, hipStream_t stream);
}  // namespace ge2e
