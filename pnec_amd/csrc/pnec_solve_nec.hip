// residual family nec (pnec_hip_mode 0)
#define PNEC_SOLVE_MODE 0
#include "pnec_solve_launch.inl"
