// pnec_solve_launch.inl -- included by one translation unit per residual family
// (PNEC_SOLVE_MODE defined by the includer) so the four families compile in parallel.
#include <cstdlib>

#include "pnec_solve_kernel.hpp"
#include "pnec_solve_group_kernel.hpp"

namespace pnec_hip {

#define PNEC_CAT_(a, b) a##b
#define PNEC_CAT(a, b) PNEC_CAT_(a, b)

#ifndef PNEC_SOLVE_AOS
hipError_t PNEC_CAT(launch_solve_mode_, PNEC_SOLVE_MODE)(int cpl, int wpp, int ldsk, bool resident,
                                                         const SolveArgs &args,
                                                         hipStream_t stream) {
  constexpr int MODE = PNEC_SOLVE_MODE;
  const dim3 grid((unsigned)args.n_solves);
  if (!resident) {
    // streaming fallback for pairs larger than any register-resident geometry
    hipLaunchKernelGGL((lm_solve_kernel<MODE, 1, kStreamWaves, 0, false>), grid, dim3(kWave * kStreamWaves), 0, stream,
                       args);
    return hipGetLastError();
  }
  // A/B form, built only with -DPNEC_SOLVE_DUAL_AB (tools/build_variant.sh) and then switched on by PNEC_SOLVE_DUAL=1: the
  // (8, 1, 3) geometry with two solves per block sharing one LM step (SRC_DUAL).  Measured in round 5 on the benchmark's
  // batch, bit-identical results: 36.68 -> 33.45 M solves/s (-8.8 %) -- the two block barriers per iteration cost more
  // than the ~16 % of issue slots the shared step saves (NOTES/round-5.md).  Not in the default library.
#ifdef PNEC_SOLVE_DUAL_AB
  static const bool dual = [] {
    const char *ev = std::getenv("PNEC_SOLVE_DUAL");
    return ev && *ev && std::atoi(ev) != 0;
  }();
  if (dual && cpl == 8 && wpp == 1 && ldsk == 3 && !args.trace) {
    if constexpr (geometry_ok(MODE, 8, 1, 3)) {
      hipLaunchKernelGGL((lm_solve_kernel<MODE, 8, 1, 3, true, SRC_DUAL>), dim3((unsigned)((args.n_solves + 1) / 2)),
                         dim3(2 * kWave), 0, stream, args);
      return hipGetLastError();
    }
  }
#endif
#define PNEC_LAUNCH_CASE(CPL, WPP, LDSK)                                                          \
  if (cpl == CPL && wpp == WPP && ldsk == LDSK) {                                                   \
    if constexpr (geometry_ok(MODE, CPL, WPP, LDSK)) {                                              \
      hipLaunchKernelGGL((lm_solve_kernel<MODE, CPL, WPP, LDSK, true>), grid, dim3(kWave * WPP), 0, \
                         stream, args);                                                             \
      return hipGetLastError();                                                                     \
    } else {                                                                                        \
      return hipErrorInvalidConfiguration;                                                          \
    }                                                                                               \
  }
  PNEC_FOR_EACH_GEOMETRY(PNEC_LAUNCH_CASE)
#undef PNEC_LAUNCH_CASE
  return hipErrorInvalidConfiguration;
}
#endif

// The multi-hypothesis form (pnec_solve_group_kernel.hpp): one block per (pair, group of WPP hypotheses).  The
// several-wavefront geometries of the ladders; bit-identical to the one-solve-per-block launch above.
#ifndef PNEC_SOLVE_AOS
hipError_t PNEC_CAT(launch_solve_group_mode_, PNEC_SOLVE_MODE)(int cpl, int wpp, int ldsk, const SolveArgs &args,
                                                               hipStream_t stream) {
  constexpr int MODE = PNEC_SOLVE_MODE;
  if (args.n_hyp < 1) return hipErrorInvalidConfiguration;
  if (wpp == 1) {   // one wavefront per pair: two hypotheses of the pair per wavefront (lm_solve_pairhyp_kernel)
    const int64_t groups1 = (args.n_hyp + kPairHyp - 1) / kPairHyp;
    const int64_t blocks1 = (args.n_solves / args.n_hyp) * groups1;
    if (blocks1 > 0x7fffffffLL) return hipErrorInvalidConfiguration;
#define PNEC_LAUNCH_CASE(CPL, WPP, LDSK)                                                                  \
  if (cpl == CPL && ldsk == LDSK) {                                                                         \
    if constexpr (pairhyp_geometry_ok(MODE, CPL, 1, LDSK)) {                                                \
      hipLaunchKernelGGL((lm_solve_pairhyp_kernel<MODE, CPL, LDSK>), dim3((unsigned)blocks1), dim3(kWave), 0, stream, args); \
      return hipGetLastError();                                                                             \
    } else {                                                                                                \
      return hipErrorInvalidConfiguration;                                                                  \
    }                                                                                                       \
  }
    PNEC_FOR_EACH_PAIRHYP_GEOMETRY(PNEC_LAUNCH_CASE)
#undef PNEC_LAUNCH_CASE
    return hipErrorInvalidConfiguration;
  }
  const int64_t groups = (args.n_hyp + wpp - 1) / wpp;
  const int64_t blocks = (args.n_solves / args.n_hyp) * groups;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  const dim3 grid((unsigned)blocks);
#define PNEC_LAUNCH_CASE(CPL, WPP, LDSK)                                                               \
  if (cpl == CPL && wpp == WPP && ldsk == LDSK) {                                                        \
    if constexpr (group_geometry_ok(MODE, CPL, WPP, LDSK)) {                                             \
      hipLaunchKernelGGL((lm_solve_group_kernel<MODE, CPL, WPP, LDSK>), grid, dim3(kWave * WPP), 0, stream, args); \
      return hipGetLastError();                                                                          \
    } else {                                                                                             \
      return hipErrorInvalidConfiguration;                                                               \
    }                                                                                                    \
  }
  PNEC_FOR_EACH_GROUP_GEOMETRY(PNEC_LAUNCH_CASE)
#undef PNEC_LAUNCH_CASE
  return hipErrorInvalidConfiguration;
}
#endif

// The AoS-source twin (streaming handle): the register-resident geometries of the auto-tuner's ladders.
#ifdef PNEC_SOLVE_AOS
hipError_t PNEC_CAT(launch_solve_aos_mode_, PNEC_SOLVE_MODE)(int cpl, int wpp, int ldsk, const SolveArgs &args,
                                                             hipStream_t stream) {
  constexpr int MODE = PNEC_SOLVE_MODE;
  const dim3 grid((unsigned)args.n_solves);
#define PNEC_LAUNCH_CASE(CPL, WPP, LDSK)                                                                    \
  if (cpl == CPL && wpp == WPP && ldsk == LDSK) {                                                             \
    if constexpr (geometry_ok(MODE, CPL, WPP, LDSK)) {                                                        \
      hipLaunchKernelGGL((lm_solve_kernel<MODE, CPL, WPP, LDSK, true, SRC_AOS>), grid, dim3(kWave * WPP), 0,  \
                         stream, args);                                                                       \
      return hipGetLastError();                                                                               \
    } else {                                                                                                  \
      return hipErrorInvalidConfiguration;                                                                    \
    }                                                                                                         \
  }
  PNEC_FOR_EACH_AOS_GEOMETRY(PNEC_LAUNCH_CASE)
#undef PNEC_LAUNCH_CASE
  return hipErrorInvalidConfiguration;
}
#endif

}  // namespace pnec_hip
