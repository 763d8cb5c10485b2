// AoS-source solve kernels of the streaming handle, residual family sym (pnec_hip_mode 3)
#define PNEC_SOLVE_MODE 3
#define PNEC_SOLVE_AOS 1
#include "pnec_solve_launch.inl"
