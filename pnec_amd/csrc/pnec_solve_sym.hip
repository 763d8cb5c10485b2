// residual family sym (pnec_hip_mode 3)
#define PNEC_SOLVE_MODE 3
#include "pnec_solve_launch.inl"
