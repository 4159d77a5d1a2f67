#pragma once
#include "ge2e_common.hpp"
namespace ge2e {
hipError_t launch_selftest_split(const float* A, const float* Bm, const float* G, float* X, float* GE, float* GC,
                                 hipStream_t stream);
hipError_t launch_selftest_wave(const float* x, float* out, hipStream_t stream);
hipError_t launch_selftest_rows16(const float* CH, const float* R, float* XT, float* GE, float* GT, hipStream_t stream);
size_t selftest_team_bytes(int payload);
hipError_t launch_selftest_team(void* ws, size_t ws_bytes, int grid, int rounds, int payload, unsigned* out,
                                hipStream_t stream);
}  // namespace ge2e
