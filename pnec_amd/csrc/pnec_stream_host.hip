// AoS-source solve kernels of the streaming handle, residual family host (pnec_hip_mode 2)
#define PNEC_SOLVE_MODE 2
#define PNEC_SOLVE_AOS 1
#include "pnec_solve_launch.inl"
