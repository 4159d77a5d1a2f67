// Launcher of GE2E_IMPL_WAVE (ge2e_wave.hip): one wave per batch, registers only, no workspace.
#pragma once
#include "ge2e_common.hpp"

namespace ge2e {

bool wave_supports(int N, int M, int D);
bool wave_is_large(int N, int M);   // more than ~30 rows: the one-wave-per-SIMD instantiation
bool wave_supports_raw(int N, int M, int D);   // ge2e_loss_fwd_bwd_raw: the register-only instantiations
hipError_t launch_wave(const Problem& p, hipStream_t stream);

}  // namespace ge2e
