// Hardware probe: 64-bit DPP on gfx950 (DP-ALU DPP supports row_newbcast only): v_fmac_f64_dpp / v_mov_b64_dpp with
// row_newbcast:N read lane N of each 16-lane row.  hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ __forceinline__ void fmac_bc(double &acc, double bc, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(y), "n"(N));
}
template <int N>
__device__ __forceinline__ void fnmac_bc(double &acc, double bc, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bc), "v"(y), "n"(N));
}
template <int N>
__device__ __forceinline__ double bcast(double x) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(N));
  return r;
}
__global__ void k(double *o) {
  const int lane = threadIdx.x;
  double x = 100.0 + lane, y = 0.5 * lane + 1.0;
  double a = 1000.0;
  fmac_bc<3>(a, x, y);          // 1000 + x[row*16+3] * y
  double b = 2000.0;
  fnmac_bc<5>(b, x, y);         // 2000 - x[row*16+5] * y
  double c = bcast<7>(x);       // x[row*16+7]
  // with a partial exec: only lanes < 40 active; source lane inside / outside the active set
  double d = -1.0, e = -1.0;
  if (lane < 40) { d = bcast<2>(x); e = bcast<9>(x); }
  o[lane * 5 + 0] = a; o[lane * 5 + 1] = b; o[lane * 5 + 2] = c; o[lane * 5 + 3] = d; o[lane * 5 + 4] = e;
}
int main() {
  double *d; hipMalloc(&d, 64 * 5 * 8);
  k<<<1, 64>>>(d);
  double h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int row = l / 16; double y = 0.5 * l + 1.0;
    double ea = 1000.0 + (100.0 + row * 16 + 3) * y, eb = 2000.0 - (100.0 + row * 16 + 5) * y, ec = 100.0 + row * 16 + 7;
    if (h[l*5] != ea || h[l*5+1] != eb || h[l*5+2] != ec) { ++bad; if (bad < 5) printf("lane %d: %g %g %g want %g %g %g\n", l, h[l*5], h[l*5+1], h[l*5+2], ea, eb, ec); }
  }
  printf("bad=%d\n", bad);
  for (int l : {0, 20, 33, 39, 40, 47, 50}) printf("lane %d partial-exec: d=%g e=%g\n", l, h[l*5+3], h[l*5+4]);
  return bad;
}
