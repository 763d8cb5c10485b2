// pnec_solve_launch.inl -- included by one translation unit per residual family
// (PNEC_SOLVE_MODE defined by the includer) so the four families compile in parallel.
#include "pnec_solve_kernel.hpp"

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
