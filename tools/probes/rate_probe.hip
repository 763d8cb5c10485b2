// Issue cost of v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 against v_fma_f64 on gfx950: N independent chains per lane, one
// wavefront alone on its SIMD and two wavefronts sharing one; wall clocks (s_memtime, 100 MHz) per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/rate_probe.hip -o gpurun_out/rate_probe && gpurun_out/rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void probe(double *out, unsigned long long *clk, int iters) {
  double x[8];
  for (int i = 0; i < 8; ++i) x[i] = 1.0 + 0.001 * (threadIdx.x + 64 * i);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) x[i] = __builtin_fma(x[i], 0.999999, 1e-9);
      if (OP == 1) x[i] = __builtin_amdgcn_rcp(x[i]);
      if (OP == 2) x[i] = __builtin_amdgcn_rsq(x[i]);
      if (OP == 3) x[i] = __builtin_amdgcn_sqrt(x[i]);
      if (OP == 4) x[i] = x[i] * 0.999999;
      if (OP == 5) x[i] = x[i] + 1e-9;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
  double *out; unsigned long long *clk;
  hipMalloc(&out, 8 * 4096 * 256); hipMalloc(&clk, 8 * 4096);
  const char *names[6] = {"v_fma_f64", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_mul_f64", "v_add_f64"};
  const int iters = 4096;
  for (int threads : {64, 128, 256, 512}) {  // 1 wave per block; blocks of 64/128/256/512 threads = 1/2/4/8 waves per CU-block
    for (int op = 0; op < 6; ++op) {
      unsigned long long h[1];
      auto launch = [&](int o) {
        switch (o) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
          case 4: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
          case 5: hipLaunchKernelGGL(probe<5>, dim3(1), dim3(threads), 0, 0, out, clk, iters); break;
        }
      };
      launch(op); launch(op);
      hipDeviceSynchronize();
      hipMemcpy(h, clk, 8, hipMemcpyDeviceToHost);
      std::printf("%d waves in one block (%d per SIMD): %-11s %.2f clocks per instruction and wavefront\n", threads / 64,
                  (threads / 64 + 3) / 4, names[op], (double)h[0] / (8.0 * iters));
    }
  }
  return 0;
}
