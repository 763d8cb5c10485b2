// pnec_front_shared.hpp -- constants the front stages' kernels (pnec_frontend.hip) and the ABI layer (pnec_capi.hip) share.
#pragma once
namespace pnec_hip {
// eigensolver schemes 1, 2: most rounds of the weighted stage (weighted_iterations - 1) whose minimisers the front
// scratch holds per pair
constexpr int kEsMaxRounds = 15;
// every allocation of a batch's planes carries this many doubles of slack behind them: the weighted stage loads its tables
// in 128-correspondence sets (16-byte loads), and the last set of a pair may read 64 doubles past the pair's last plane
constexpr int kDataSlackDoubles = 64;
// the front stages' scratch of a batch of P pairs: doubles and ints per pair, plus a few ints of counters behind the ints
// (front_scratch in pnec_frontend.hip lays them out, pnec_capi.hip allocates them)
constexpr int kFrontDoublesPerPair = 43 + 3 * kEsMaxRounds;
constexpr int kFrontIntsPerPair = 4;
constexpr int kFrontCounterInts = 16;
}  // namespace pnec_hip
