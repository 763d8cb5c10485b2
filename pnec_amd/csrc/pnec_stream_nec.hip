// AoS-source solve kernels of the streaming handle, residual family nec (pnec_hip_mode 0)
#define PNEC_SOLVE_MODE 0
#define PNEC_SOLVE_AOS 1
#include "pnec_solve_launch.inl"
