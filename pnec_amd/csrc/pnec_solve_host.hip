// residual family host (pnec_hip_mode 2)
#define PNEC_SOLVE_MODE 2
#include "pnec_solve_launch.inl"
