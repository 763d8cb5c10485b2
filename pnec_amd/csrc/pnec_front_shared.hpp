// pnec_front_shared.hpp -- constants the front stages' kernels (pnec_frontend.hip) and the ABI layer (pnec_capi.hip) share.
#pragma once
namespace pnec_hip {
// eigensolver schemes 1, 2: most rounds of the weighted stage (weighted_iterations - 1) whose minimisers the front
// scratch holds per pair
constexpr int kEsMaxRounds = 15;
}  // namespace pnec_hip
