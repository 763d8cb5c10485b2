// AoS-source solve kernels of the streaming handle, residual family target (pnec_hip_mode 1)
#define PNEC_SOLVE_MODE 1
#define PNEC_SOLVE_AOS 1
#include "pnec_solve_launch.inl"
