// pnec_device.hpp -- device-side building blocks of the MI355X PNEC solver (gfx950 only).
//
// Everything here is wave64 code for CDNA4: one wavefront (or a few) owns one solve, keeps the
// pair's bearings/covariances in registers for the whole Levenberg-Marquardt loop, and reduces the
// 21 normal-equation sums with DPP / v_permlane*_swap cross-lane moves (plus one LDS hop when
// several wavefronts share a solve).  No MFMA: the contraction is 5x5.
//
// Maths follows the reference functors (include/optimization/pnec_residual.h:50-150,
// nec_residual.h:47-68) in the factored form of SURVEY.md Appendix A:
//   m = t x f1,  g = R' m,  n = f2.g,  r = n / sqrt(g' S g + reg)        (TARGET)
//   J_omega = (R dr/dg) x m,   J_t = f1 x (R dr/dg),   delta = omega / 2 (EigenQuaternionManifold)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pnec_hip.h"

namespace pnec_hip {

constexpr int kWave = 64;
constexpr int kNumAcc = 21;  // sum r^2 | J'r (5) | upper triangle of J'J (15)

__host__ __device__ constexpr int num_components(int mode) {
  return mode == PNEC_HIP_MODE_NEC ? 6 : (mode == PNEC_HIP_MODE_SYM ? 18 : 12);
}
// packed upper-triangular index of a symmetric 5x5, a <= b
__host__ __device__ constexpr int tri(int a, int b) { return a * 5 - a * (a - 1) / 2 + (b - a); }
__host__ __device__ constexpr int sym(int a, int b) { return a <= b ? tri(a, b) : tri(b, a); }

// ------------------------------------------------------------------------------------------
// scalar helpers
__device__ __forceinline__ double make_double(int hi, int lo) { return __hiloint2double(hi, lo); }

// Move a wave-uniform double into scalar registers (2 x v_readfirstlane_b32).
__device__ __forceinline__ double to_sgpr(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return make_double(hi, lo);
}
__device__ __forceinline__ int to_sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }

// 1/sqrt(x): v_rsq_f64 seed (~2^-23 rel.) + one third-order correction -> < 1 ulp-ish.
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-(x * y0), y0, 1.0);
  return __builtin_fma(y0 * e, __builtin_fma(e, 0.375, 0.5), y0);
}
// 1/x: v_rcp_f64 seed + two Newton steps.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}
__device__ __forceinline__ bool finite_d(double x) { return __builtin_isfinite(x); }
// sqrt(x) for x >= 0 via the reciprocal square root (0 stays 0)
__device__ __forceinline__ double fast_sqrt(double x) { return x > 0.0 ? x * fast_rsqrt(x) : 0.0; }

// sin and cos together, for the angles an LM run produces (|x| well below 1e6): two-constant
// FMA Cody-Waite reduction to [-pi/4, pi/4] + the classic degree-13/14 minimax kernels.
// ~35 VALU instructions instead of the generic libm path with its Payne-Hanek branch.
__device__ __forceinline__ void sincos_bounded(double x, double &s, double &c) {
  const double k = __builtin_rint(x * 0.63661977236758134308);  // 2/pi
  double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(r * z, ps, r);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int q = (int)k & 3;
  const double a = (q & 1) ? cr : sr;
  const double b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// The same with every constant pinned to the spot where it is used (a scalar register pair through an empty volatile asm
// statement: two s_mov per constant and call).  Left to itself the compiler materialises the thirteen coefficients ONCE,
// in vector registers outside the solve's loop, and in the geometries whose loop has no register to spare -- the tail form
// (12, 1, 3), the multi-wavefront ones -- it spills them to scratch and reloads them in every call of the LM step: twenty
// memory round trips on its latency chain (ISA marks, round 4).
__device__ __forceinline__ double pinned_const(double c) {
  asm volatile("" : "+s"(c));
  return c;
}
__device__ __forceinline__ void sincos_bounded_pinned(double x, double &s, double &c) {
  const double k = __builtin_rint(x * pinned_const(0.63661977236758134308));  // 2/pi
  double r = __builtin_fma(-k, pinned_const(1.57079632679489655800e+00), x);
  r = __builtin_fma(-k, pinned_const(6.12323399573676603587e-17), r);
  const double z = r * r;
  double ps = __builtin_fma(z, pinned_const(1.58969099521155010221e-10), pinned_const(-2.50507602534068634195e-08));
  ps = __builtin_fma(z, ps, pinned_const(2.75573137070700676789e-06));
  ps = __builtin_fma(z, ps, pinned_const(-1.98412698298579493134e-04));
  ps = __builtin_fma(z, ps, pinned_const(8.33333333332248946124e-03));
  ps = __builtin_fma(z, ps, pinned_const(-1.66666666666666324348e-01));
  const double sr = __builtin_fma(r * z, ps, r);
  double pc = __builtin_fma(z, pinned_const(-1.13596475577881948265e-11), pinned_const(2.08757232129817482790e-09));
  pc = __builtin_fma(z, pc, pinned_const(-2.75573143513906633035e-07));
  pc = __builtin_fma(z, pc, pinned_const(2.48015872894767294178e-05));
  pc = __builtin_fma(z, pc, pinned_const(-1.38888888888741095749e-03));
  pc = __builtin_fma(z, pc, pinned_const(4.16666666666666019037e-02));
  const double cr = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
  const int q = (int)k & 3;
  const double a = (q & 1) ? cr : sr;
  const double b = (q & 1) ? sr : cr;
  s = (q & 2) ? -a : a;
  c = ((q + 1) & 2) ? -b : b;
}

// atan / atan2 / acos for the start angles of a solve (AnglesFromVec, once per solve in one lane):
// the classic argument reduction at 7/16, 11/16, 19/16, 39/16 with an 11-term odd polynomial
// (<= 1 ulp of libm, self-tested on the device) -- ~60 instructions instead of the generic libm
// entry points with their special-case ladders.
__device__ __forceinline__ double atan_lean(double x) {
  const double ax = fabs(x);
  double hi = 0.0, lo = 0.0, r = ax;
  if (ax >= 0.4375) {
    if (ax < 0.6875) {
      hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17;
      r = (2.0 * ax - 1.0) / (2.0 + ax);
    } else if (ax < 1.1875) {
      hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17;
      r = (ax - 1.0) / (ax + 1.0);
    } else if (ax < 2.4375) {
      hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17;
      r = (ax - 1.5) / (1.0 + 1.5 * ax);
    } else {
      hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17;
      r = -1.0 / ax;
    }
  }
  const double z = r * r, w = z * z;
  double s1 = __builtin_fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
  s1 = __builtin_fma(w, s1, 6.66107313738753120669e-02);
  s1 = __builtin_fma(w, s1, 9.09088713343650656196e-02);
  s1 = __builtin_fma(w, s1, 1.42857142725034663711e-01);
  s1 = z * __builtin_fma(w, s1, 3.33333333333329318027e-01);
  double s2 = __builtin_fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
  s2 = __builtin_fma(w, s2, -7.69187620504482999495e-02);
  s2 = __builtin_fma(w, s2, -1.11111104054623557880e-01);
  s2 = w * __builtin_fma(w, s2, -1.99999999998764832476e-01);
  const double res = (ax < 0.4375) ? r - r * (s1 + s2) : hi - ((r * (s1 + s2) - lo) - r);
  return x < 0.0 ? -res : res;
}
// atan2 for finite arguments (the callers pass components of a unit vector)
__device__ __forceinline__ double atan2_lean(double y, double x) {
  const double kPi = 3.14159265358979311600e+00, kPiLo = 1.2246467991473531772e-16;
  if (x == 0.0 && y == 0.0) return 0.0;  // the +-0 / pi cases of libm: sign of x (callers never need -0)
  if (x == 0.0) return y > 0.0 ? 0.5 * kPi : -0.5 * kPi;
  const double a = atan_lean(fabs(y / x));
  double res = x > 0.0 ? a : kPi - (a - kPiLo);
  return y < 0.0 ? -res : res;
}
// acos(c), |c| <= 1: 2 atan2(sqrt(1 - c), sqrt(1 + c)) -- accurate at both ends of the range
__device__ __forceinline__ double acos_lean(double c) {
  return 2.0 * atan2_lean(sqrt(1.0 - c), sqrt(1.0 + c));
}

// ------------------------------------------------------------------------------------------
// cross-lane sum over the 64 lanes of a wavefront; every lane ends with the same bits.
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  // every lane is written (full masks, in-row permutations): no "old" value to preserve, so the
  // compiler need not copy the source first
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return make_double(hi, lo);
}
// value of lane K of the caller's quad (all four lanes of the quad must be active)
template <int K>
__device__ __forceinline__ double quad_broadcast(double x) {
  return dpp_perm<K | (K << 2) | (K << 4) | (K << 6)>(x);
}
template <int K>
__device__ __forceinline__ int quad_broadcast(int x) {
  return __builtin_amdgcn_mov_dpp(x, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true);
}
// 64-bit DPP.  The DP ALU of gfx90a / gfx94x / gfx950 supports exactly one DPP control, row_newbcast:N -- "every lane
// of a 16-lane row reads lane N of its row" -- on its VOP1 / VOP2 encodings (v_mov_b64, v_fmac_f64, ...).  That is a
// 64-bit broadcast FUSED into the consuming multiply-add: no separate cross-lane instruction (a 64-bit value moved
// with the 32-bit DPP moves costs two VALU slots).  The compiler has no builtin that selects these (its update_dpp
// builtin is 32-bit), so they are spelled in assembly; "s_nop 1" in front covers the DPP read-after-VALU-write hazard
// (2 wait states) the hazard recogniser cannot see through inline assembly.  The source lane must be active.
// (tools/dpp_probe.hip is the hardware probe of these semantics; pnec_hip_selftest repeats it.)
//   acc += src[lane N of this row] * y
template <int N>
__device__ __forceinline__ void fmac_row(double &acc, double src, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(src), "v"(y), "n"(N));
}
//   acc -= src[lane N of this row] * y
template <int N>
__device__ __forceinline__ void fnmac_row(double &acc, double src, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc) : "v"(src), "v"(y), "n"(N));
}
//   src[lane N of this row]
template <int N>
__device__ __forceinline__ double bcast_row(double src) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(N));
  return r;
}

// lane ^ XOR inside a group of 32 through the LDS crossbar (ds_swizzle_b32, bit-mask mode): the
// exchange runs on the LDS pipe, not the VALU the solver is bound by
template <int XOR>
__device__ __forceinline__ double swizzle_xor(double x) {
  constexpr int pattern = (XOR << 10) | 0x1F;
  const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), pattern);
  const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), pattern);
  return make_double(hi, lo);
}
// rows {1,3} of one copy <-> rows {0,2} of the other: sum = pairwise row sums in all 4 rows
__device__ __forceinline__ double row_pair_sum(double x) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return make_double((int)b[0], (int)a[0]) + make_double((int)b[1], (int)a[1]);
}
// lanes 32..63 of one copy <-> lanes 0..31 of the other
__device__ __forceinline__ double half_pair_sum(double x) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return make_double((int)b[0], (int)a[0]) + make_double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double wave_allreduce_sum(double x) {
  x += dpp_perm<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_perm<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_perm<0x141>(x);  // row_half_mirror
  x += dpp_perm<0x140>(x);  // row_mirror
  x = row_pair_sum(x);      // v_permlane16_swap (gfx950)
  x = half_pair_sum(x);     // v_permlane32_swap (gfx950)
  return x;
}


// Sum each of the 21 accumulators over the 64 lanes and return the sums in scalar registers.
// v_permlane32_swap / v_permlane16_swap exchange half of one register with the other half of
// a second one, so one swap + one add reduces TWO accumulators at once with no selects:
//   swap32(X, Y): X' = [X.lo | Y.lo], Y' = [X.hi | Y.hi]  =>  X' + Y' = [X.lo+X.hi | Y.lo+Y.hi]
// 21 -> 11 values (lane halves) -> 6 values (rows of 16 lanes); the last four levels run as a
// butterfly inside each row on those 6 values (row_allreduce_sum); the four row
// leaders then store their sums to LDS, from where every lane reads all 21 back (broadcast).
__device__ __forceinline__ double swap_add32(double x, double y) {
  const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
  const unsigned yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
  const auto a = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
  return make_double((int)b[0], (int)a[0]) + make_double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double swap_add16(double x, double y) {
  const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x);
  const unsigned yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
  const auto a = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
  return make_double((int)b[0], (int)a[0]) + make_double((int)b[1], (int)a[1]);
}
// single-precision twins (the direction pre-screen of the weighted stage)
__device__ __forceinline__ float swap_add32_f(float x, float y) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
__device__ __forceinline__ float swap_add16_f(float x, float y) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_perm_f(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_allreduce_sum_f(float x) {
  x += dpp_perm_f<0xB1>(x);
  x += dpp_perm_f<0x4E>(x);
  x += dpp_perm_f<0x141>(x);
  x += dpp_perm_f<0x140>(x);
  return x;
}
__device__ __forceinline__ float wave_allreduce_min_f(float x) {
  x = fminf(x, dpp_perm_f<0xB1>(x));
  x = fminf(x, dpp_perm_f<0x4E>(x));
  x = fminf(x, dpp_perm_f<0x141>(x));
  x = fminf(x, dpp_perm_f<0x140>(x));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fminf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fminf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// Sum over the 16 lanes of each DPP row.  Two forms: DPP moves (VALU: 2 moves + 1 add per level)
// or the LDS crossbar (ds_swizzle: 1 add per level on the VALU, but an LDS round trip of latency
// per level).  Measured on the solver: with the LM step's latency chain in
// the same wavefront, the four extra round trips cost more than the 48 VALU slots they save.
__device__ __forceinline__ double row_allreduce_sum(double x) {
#ifdef PNEC_ROW_SWIZZLE
  x += swizzle_xor<1>(x);
  x += swizzle_xor<2>(x);
  x += swizzle_xor<4>(x);
  x += swizzle_xor<8>(x);
#else
  x += dpp_perm<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_perm<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_perm<0x141>(x);  // row_half_mirror
  x += dpp_perm<0x140>(x);  // row_mirror
#endif
  return x;
}
template <int LANE>
__device__ __forceinline__ double read_lane(double x) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), LANE);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), LANE);
  return make_double(hi, lo);
}
// Where sum j lands in the 24-slot table the row leaders write (row r owns slots 6r .. 6r+5):
// row 0: acc[0..5], row 1: acc[6..10], row 2: acc[11..16], row 3: acc[17..20].
constexpr int kSumSlots = 24;
__host__ __device__ constexpr int sum_slot(int j) {
  return j < 6 ? j : (j < 11 ? 6 + (j - 6) : (j < 17 ? 12 + (j - 11) : 18 + (j - 17)));
}
// 21 accumulators -> 6 registers whose 16-lane rows each hold the complete sums that row owns
// (every lane of the row has the same bits).
__device__ __forceinline__ void wave_reduce21_rows(const double (&acc)[kNumAcc], double (&c)[6]) {
  // level 1 (lanes l <-> l+32): b[i] holds acc[i] in lanes 0..31 and acc[i+11] in lanes 32..63
  double b[11];
#pragma unroll
  for (int i = 0; i < 10; ++i) b[i] = swap_add32(acc[i], acc[i + 11]);
  b[10] = swap_add32(acc[10], 0.0);
  // level 2 (rows r <-> r^1): c[i] rows {0,2} hold b[i], rows {1,3} hold b[i+6]
#pragma unroll
  for (int i = 0; i < 5; ++i) c[i] = swap_add16(b[i], b[i + 6]);
  c[5] = swap_add16(b[5], 0.0);
  // levels 3..6 inside each row
#pragma unroll
  for (int i = 0; i < 6; ++i) c[i] = row_allreduce_sum(c[i]);
}
// accumulator 0 alone, through the same tree (same additions in the same order as in
// wave_reduce21_rows, so the sum has the same bits): valid in the 16 lanes of row 0
__device__ __forceinline__ double wave_reduce_acc0_row0(double acc0) {
  const double b0 = swap_add32(acc0, 0.0);
  const double c0 = swap_add16(b0, 0.0);
  return row_allreduce_sum(c0);
}
// the same sums as wave-uniform values, picked from the owning rows with v_readlane
__device__ __forceinline__ void wave_reduce21(const double (&acc)[kNumAcc], double (&sum)[kNumAcc]) {
  double c[6];
  wave_reduce21_rows(acc, c);
#pragma unroll
  for (int i = 0; i < 6; ++i) sum[i] = read_lane<0>(c[i]);
#pragma unroll
  for (int i = 0; i < 5; ++i) sum[i + 6] = read_lane<16>(c[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) sum[i + 11] = read_lane<32>(c[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) sum[i + 17] = read_lane<48>(c[i]);
}
// true when no lane of the wavefront holds an Inf/NaN in any of the six registers
__device__ __forceinline__ bool rows_all_finite(const double (&c)[6]) {
  double z = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) z = __builtin_fma(c[i], 0.0, z);  // 0 for finite, NaN otherwise
  return __builtin_amdgcn_ballot_w64(!(z == 0.0)) == 0ull;
}

// One correspondence per lane from a pair's SoA planes in memory: plane c of the pair starts at sbase + c * plane_bytes
// (scalar registers), the lane's correspondence sits `lo` + IMM bytes into it.  (The solver's tail pass, geometry
// (12, 1, 3).)  Spelled as the scalar-base form of global_load (saddr + 32-bit voffset + immediate) because the
// compiler, left to itself, keeps a 64-bit vector address per plane and slot alive across its loops -- registers these
// kernels do not have: the solver spilled them, the weighted stage reloaded 96 of them from scratch one by one, each in
// front of its load.  Lanes beyond the planes (`in` false) keep the zeros they came with.
//
// ONE asm statement issues the loads AND waits for them (round 5; until then a statement per load and a separate wait
// statement).  To the compiler an asm statement has written its outputs when it returns: between an issue statement and
// a later wait statement it is free to copy, spill or re-use a destination register while the load that will really
// write it is still in flight -- it did, once, under a tighter register budget (tools/check_asm_loads.py caught it after
// the link; that check stays as a backstop, but correctness no longer rests on it).  Inside one statement there is no
// "between".  The s_nop covers a plane base the compiler has just fetched with a VALU instruction (v_readlane from a
// spilled-SGPR lane, v_readfirstlane): a vector-memory instruction reading that SGPR needs five wait states (CDNA ISA,
// "manually inserted wait states"), which the compiler inserts for its own instructions and cannot see into this text.
#define PNEC_LD_(D, B, OFF) "global_load_dwordx2 %[" #D "], %[lo], %[" #B "] offset:" OFF "\n\t"
template <int NC, int IMM>
__device__ __forceinline__ void load_planes_saddr(double (&e)[NC], const char *sbase, size_t plane_bytes, unsigned lo, bool in) {
  static_assert(NC == 6 || NC == 12, "tail form: 6- and 12-plane payloads");
#pragma unroll
  for (int c = 0; c < NC; ++c) e[c] = 0.0;
  if (in) {
    if constexpr (NC == 12) {
      asm volatile("s_nop 4\n\t"
                   PNEC_LD_(d0, b0, "%[imm]") PNEC_LD_(d1, b1, "%[imm]") PNEC_LD_(d2, b2, "%[imm]") PNEC_LD_(d3, b3, "%[imm]")
                   PNEC_LD_(d4, b4, "%[imm]") PNEC_LD_(d5, b5, "%[imm]") PNEC_LD_(d6, b6, "%[imm]") PNEC_LD_(d7, b7, "%[imm]")
                   PNEC_LD_(d8, b8, "%[imm]") PNEC_LD_(d9, b9, "%[imm]") PNEC_LD_(d10, b10, "%[imm]") PNEC_LD_(d11, b11, "%[imm]")
                   "s_waitcnt vmcnt(0)"
                   : [d0] "+v"(e[0]), [d1] "+v"(e[1]), [d2] "+v"(e[2]), [d3] "+v"(e[3]), [d4] "+v"(e[4]), [d5] "+v"(e[5]),
                     [d6] "+v"(e[6]), [d7] "+v"(e[7]), [d8] "+v"(e[8]), [d9] "+v"(e[9]), [d10] "+v"(e[10]), [d11] "+v"(e[11])
                   : [lo] "v"(lo), [imm] "n"(IMM), [b0] "s"(sbase), [b1] "s"(sbase + plane_bytes), [b2] "s"(sbase + 2 * plane_bytes),
                     [b3] "s"(sbase + 3 * plane_bytes), [b4] "s"(sbase + 4 * plane_bytes), [b5] "s"(sbase + 5 * plane_bytes),
                     [b6] "s"(sbase + 6 * plane_bytes), [b7] "s"(sbase + 7 * plane_bytes), [b8] "s"(sbase + 8 * plane_bytes),
                     [b9] "s"(sbase + 9 * plane_bytes), [b10] "s"(sbase + 10 * plane_bytes), [b11] "s"(sbase + 11 * plane_bytes)
                   : "memory");
    } else {
      asm volatile("s_nop 4\n\t"
                   PNEC_LD_(d0, b0, "%[imm]") PNEC_LD_(d1, b1, "%[imm]") PNEC_LD_(d2, b2, "%[imm]") PNEC_LD_(d3, b3, "%[imm]")
                   PNEC_LD_(d4, b4, "%[imm]") PNEC_LD_(d5, b5, "%[imm]")
                   "s_waitcnt vmcnt(0)"
                   : [d0] "+v"(e[0]), [d1] "+v"(e[1]), [d2] "+v"(e[2]), [d3] "+v"(e[3]), [d4] "+v"(e[4]), [d5] "+v"(e[5])
                   : [lo] "v"(lo), [imm] "n"(IMM), [b0] "s"(sbase), [b1] "s"(sbase + plane_bytes), [b2] "s"(sbase + 2 * plane_bytes),
                     [b3] "s"(sbase + 3 * plane_bytes), [b4] "s"(sbase + 4 * plane_bytes), [b5] "s"(sbase + 5 * plane_bytes)
                   : "memory");
    }
  }
}

// EIGHT correspondences per lane of a pair's TWELVE planes in one statement with its wait (the weighted stage's table
// build), as FOUR sets of 16-byte loads: set K = 0..3 gives the lane correspondences 128 K + 2 lane and + 1 of this
// wavefront's share (slots 2 K, 2 K + 1), 1024 K bytes into each plane -- 48 loads per lane.  (Until round 5: 96 8-byte
// loads, correspondence 64 k + lane in slot k.  A wavefront can have 63 vector-memory instructions in flight: the 64th
// waited for the first to return, so the build took two trips to memory and more -- 22 k of the pair's 80 k clocks; 48
// fit in one.  Which lane and slot holds which correspondence changes the ORDER of the stage's sums over the pair, i.e.
// its results in the last bits.)  Only the first `nt` sets are loaded -- the sets that start inside the pair's planes --
// skipped wave-uniformly; the others keep the zeros they came with.  The planes are padded with zeros to a multiple of
// 64 correspondences, a set spans 128: the last set may read 512 bytes past a plane's end -- the next plane's start, or,
// behind the last plane, whatever follows the pair's block (the next pair, or the 64 doubles of slack every batch
// allocation carries for this); what it reads there belongs to lanes beyond the pair, which the caller discards.
typedef double pnec_d2_ __attribute__((ext_vector_type(2)));
#define PNEC_LD4_(D, B, OFF) "global_load_dwordx4 %[" #D "], %[lo], %[" #B "] offset:" OFF "\n\t"
#define PNEC_SET12X2_(K, OFF)                                                                                      \
  "s_cmp_le_u32 %[nt], " #K "\n\ts_cbranch_scc1 1f\n\t"                                                            \
  PNEC_LD4_(d##K##_0, b0, OFF) PNEC_LD4_(d##K##_1, b1, OFF) PNEC_LD4_(d##K##_2, b2, OFF) PNEC_LD4_(d##K##_3, b3, OFF)   \
  PNEC_LD4_(d##K##_4, b4, OFF) PNEC_LD4_(d##K##_5, b5, OFF) PNEC_LD4_(d##K##_6, b6, OFF) PNEC_LD4_(d##K##_7, b7, OFF)   \
  PNEC_LD4_(d##K##_8, b8, OFF) PNEC_LD4_(d##K##_9, b9, OFF) PNEC_LD4_(d##K##_10, b10, OFF) PNEC_LD4_(d##K##_11, b11, OFF)
#define PNEC_SET12X2_OUT_(K)                                                                                        \
  [d##K##_0] "+v"(x[K][0]), [d##K##_1] "+v"(x[K][1]), [d##K##_2] "+v"(x[K][2]), [d##K##_3] "+v"(x[K][3]),           \
  [d##K##_4] "+v"(x[K][4]), [d##K##_5] "+v"(x[K][5]), [d##K##_6] "+v"(x[K][6]), [d##K##_7] "+v"(x[K][7]),           \
  [d##K##_8] "+v"(x[K][8]), [d##K##_9] "+v"(x[K][9]), [d##K##_10] "+v"(x[K][10]), [d##K##_11] "+v"(x[K][11])
__device__ __forceinline__ void load_sets4x12x2_saddr(double (&pe)[8][12], const char *sbase, size_t plane_bytes, unsigned lo16,
                                                      unsigned nt) {
  pnec_d2_ x[4][12];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 12; ++c) x[k][c] = pnec_d2_{0.0, 0.0};
  asm volatile("s_nop 4\n\t"
               PNEC_SET12X2_(0, "0") PNEC_SET12X2_(1, "1024") PNEC_SET12X2_(2, "2048") PNEC_SET12X2_(3, "3072")
               "1:\n\ts_waitcnt vmcnt(0)"
               : PNEC_SET12X2_OUT_(0), PNEC_SET12X2_OUT_(1), PNEC_SET12X2_OUT_(2), PNEC_SET12X2_OUT_(3)
               : [lo] "v"(lo16), [nt] "s"(nt), [b0] "s"(sbase), [b1] "s"(sbase + plane_bytes), [b2] "s"(sbase + 2 * plane_bytes),
                 [b3] "s"(sbase + 3 * plane_bytes), [b4] "s"(sbase + 4 * plane_bytes), [b5] "s"(sbase + 5 * plane_bytes),
                 [b6] "s"(sbase + 6 * plane_bytes), [b7] "s"(sbase + 7 * plane_bytes), [b8] "s"(sbase + 8 * plane_bytes),
                 [b9] "s"(sbase + 9 * plane_bytes), [b10] "s"(sbase + 10 * plane_bytes), [b11] "s"(sbase + 11 * plane_bytes)
               : "memory", "scc");
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int c = 0; c < 12; ++c) {
      pe[2 * k][c] = x[k][c].x;
      pe[2 * k + 1][c] = x[k][c].y;
    }
}

// ------------------------------------------------------------------------------------------
// pose-dependent, correspondence-independent quantities of one pass
struct PassUniforms {
  double R[9];    // row-major, Eigen's un-normalised quaternion->matrix formula
  double t[3];    // (sin th cos ph, sin th sin ph, cos th)
  double bth[3];  // d t / d theta
  double bph[3];  // d t / d phi  (z component is 0)
};

// Eigen::QuaternionBase::toRotationMatrix(), q = xyzw, no normalisation (pnec_residual.h:92-93)
__device__ __forceinline__ void rot_from_quat(const double (&q)[4], double (&R)[9]) {
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

__device__ __forceinline__ void make_uniforms(double theta, double phi, const double (&q)[4],
                                              PassUniforms &U) {
  double R[9];
  rot_from_quat(q, R);
  double st, ct, sp, cp;
  sincos_bounded(theta, st, ct);
  sincos_bounded(phi, sp, cp);
#pragma unroll
  for (int i = 0; i < 9; ++i) U.R[i] = to_sgpr(R[i]);
  U.t[0] = to_sgpr(st * cp);   U.t[1] = to_sgpr(st * sp);   U.t[2] = to_sgpr(ct);
  U.bth[0] = to_sgpr(ct * cp); U.bth[1] = to_sgpr(ct * sp); U.bth[2] = to_sgpr(-st);
  U.bph[0] = to_sgpr(-st * sp); U.bph[1] = to_sgpr(st * cp); U.bph[2] = 0.0;
}

// common.cc:103-116 (AnglesFromVec)
__device__ __forceinline__ void angles_from_vec(double x, double y, double z, double &theta,
                                                double &phi) {
  const double n = sqrt(x * x + y * y + z * z);
  if (n == 0.0) {
    theta = 0.0;
    phi = 0.0;
    return;
  }
  theta = acos_lean(z / n);
  phi = (fabs(theta) < 1e-10) ? 0.0 : atan2_lean(y / n, x / n);
}

// ------------------------------------------------------------------------------------------
// One correspondence: residual r and tangent-space Jacobian (theta, phi, omega_xyz) -- the
// omega columns are for R <- Exp(omega) R; the caller scales them by 2 (delta = omega/2) once,
// after the reduction.  d[] = this correspondence's planes:
//   0..2 f1 | 3..5 f2 | 6..11 cov (xx,xy,xz,yy,yz,zz) | 12..17 cov_host (SYM only)
// Padding slots (a lane's share beyond the pair's last correspondence) hold zeros in every plane:
// f2 = 0 makes n = 0 and dr/dg = 0, so residual and Jacobian come out exactly 0 without a per-
// correspondence select -- provided 1/sqrt(den) is finite there.  den of a padding slot is `reg`,
// which may be 0, hence the clamp below; for a real correspondence den >= kTinyDen always unless the
// input is degenerate (zero covariance AND zero regularisation), where the reference's own
// arithmetic is Inf/NaN and the solve is reported as failed either way.
constexpr double kTinyDen = 1e-300;
template <int MODE>
__device__ __forceinline__ void eval_corr(const double (&d)[num_components(MODE)],
                                          const PassUniforms &U, double reg, double &r,
                                          double (&J)[5]) {
  const double f1x = d[0], f1y = d[1], f1z = d[2];
  const double f2x = d[3], f2y = d[4], f2z = d[5];
  const double *R = U.R;
  // m = t x f1
  const double mx = U.t[1] * f1z - U.t[2] * f1y;
  const double my = U.t[2] * f1x - U.t[0] * f1z;
  const double mz = U.t[0] * f1y - U.t[1] * f1x;
  // g = R' m
  const double gx = R[0] * mx + R[3] * my + R[6] * mz;
  const double gy = R[1] * mx + R[4] * my + R[7] * mz;
  const double gz = R[2] * mx + R[5] * my + R[8] * mz;
  const double n = f2x * gx + f2y * gy + f2z * gz;

  double wx, wy, wz;          // dr/dg
  double ex = 0, ey = 0, ez = 0;  // extra J_omega term (HOST / SYM)
  double hx = 0, hy = 0, hz = 0;  // extra J_t term
  if constexpr (MODE == PNEC_HIP_MODE_NEC) {
    r = n;
    wx = f2x; wy = f2y; wz = f2z;
  } else if constexpr (MODE == PNEC_HIP_MODE_TARGET) {
    const double sgx = d[6] * gx + d[7] * gy + d[8] * gz;
    const double sgy = d[7] * gx + d[9] * gy + d[10] * gz;
    const double sgz = d[8] * gx + d[10] * gy + d[11] * gz;
    const double den = gx * sgx + gy * sgy + gz * sgz + reg;
    const double y = fast_rsqrt(fmax(den, kTinyDen));
    r = n * y;
    const double c = r * y;  // n / den
    wx = y * (f2x - c * sgx);
    wy = y * (f2y - c * sgy);
    wz = y * (f2z - c * sgz);
  } else {
    // HOST: den = h' S h + reg, h = t x (R f1), S = cov.
    // SYM : den = g' S2 g + h' S1 h + reg, h = t x (R f2), S2 = cov, S1 = cov_host.
    constexpr bool kSym = (MODE == PNEC_HIP_MODE_SYM);
    const double ax = kSym ? f2x : f1x, ay = kSym ? f2y : f1y, az = kSym ? f2z : f1z;
    const double px = R[0] * ax + R[1] * ay + R[2] * az;
    const double py = R[3] * ax + R[4] * ay + R[5] * az;
    const double pz = R[6] * ax + R[7] * ay + R[8] * az;
    const double qx = U.t[1] * pz - U.t[2] * py;  // h = t x p
    const double qy = U.t[2] * px - U.t[0] * pz;
    const double qz = U.t[0] * py - U.t[1] * px;
    constexpr int o = kSym ? 12 : 6;
    const double shx = d[o + 0] * qx + d[o + 1] * qy + d[o + 2] * qz;
    const double shy = d[o + 1] * qx + d[o + 3] * qy + d[o + 4] * qz;
    const double shz = d[o + 2] * qx + d[o + 4] * qy + d[o + 5] * qz;
    double den = qx * shx + qy * shy + qz * shz + reg;
    double sgx = 0, sgy = 0, sgz = 0;
    if constexpr (kSym) {
      sgx = d[6] * gx + d[7] * gy + d[8] * gz;
      sgy = d[7] * gx + d[9] * gy + d[10] * gz;
      sgz = d[8] * gx + d[10] * gy + d[11] * gz;
      den += gx * sgx + gy * sgy + gz * sgz;
    }
    const double y = fast_rsqrt(fmax(den, kTinyDen));
    r = n * y;
    const double c = r * y;
    wx = y * (f2x - c * sgx);
    wy = y * (f2y - c * sgy);
    wz = y * (f2z - c * sgz);
    // wh = dr/dh = -n S h / den^(3/2)
    const double k = -c * y;
    const double whx = k * shx, why = k * shy, whz = k * shz;
    // J_t += p x wh ;  J_omega += p x (wh x t)
    hx = py * whz - pz * why;
    hy = pz * whx - px * whz;
    hz = px * why - py * whx;
    const double vx = why * U.t[2] - whz * U.t[1];
    const double vy = whz * U.t[0] - whx * U.t[2];
    const double vz = whx * U.t[1] - why * U.t[0];
    ex = py * vz - pz * vy;
    ey = pz * vx - px * vz;
    ez = px * vy - py * vx;
  }
  // u = R w
  const double ux = R[0] * wx + R[1] * wy + R[2] * wz;
  const double uy = R[3] * wx + R[4] * wy + R[5] * wz;
  const double uz = R[6] * wx + R[7] * wy + R[8] * wz;
  // J_omega = u x m (+ e)
  J[2] = uy * mz - uz * my + ex;
  J[3] = uz * mx - ux * mz + ey;
  J[4] = ux * my - uy * mx + ez;
  // J_t = f1 x u (+ h), projected on the (theta, phi) chart
  const double jx = f1y * uz - f1z * uy + hx;
  const double jy = f1z * ux - f1x * uz + hy;
  const double jz = f1x * uy - f1y * ux + hz;
  J[0] = U.bth[0] * jx + U.bth[1] * jy + U.bth[2] * jz;
  J[1] = U.bph[0] * jx + U.bph[1] * jy;
}

// The residual alone (the pass whose Jacobian can never be used: the candidate evaluated at the
// iteration cap).  Same operations in the same order as eval_corr up to r, so the cost is the bits a
// full pass would have produced.  `k` is a quantity that is non-finite whenever the Jacobian would be
// (n / den^(3/2): every Jacobian entry is a bounded multiple of y and of k), 0 * k is accumulated as
// the finite-Jacobian witness Ceres' "Jacobian evaluation failed" test needs.
template <int MODE>
__device__ __forceinline__ void eval_cost(const double (&d)[num_components(MODE)],
                                          const PassUniforms &U, double reg, double &r, double &k) {
  const double f1x = d[0], f1y = d[1], f1z = d[2];
  const double f2x = d[3], f2y = d[4], f2z = d[5];
  const double *R = U.R;
  const double mx = U.t[1] * f1z - U.t[2] * f1y;
  const double my = U.t[2] * f1x - U.t[0] * f1z;
  const double mz = U.t[0] * f1y - U.t[1] * f1x;
  const double gx = R[0] * mx + R[3] * my + R[6] * mz;
  const double gy = R[1] * mx + R[4] * my + R[7] * mz;
  const double gz = R[2] * mx + R[5] * my + R[8] * mz;
  const double n = f2x * gx + f2y * gy + f2z * gz;
  if constexpr (MODE == PNEC_HIP_MODE_NEC) {
    r = n;
    k = n;
  } else if constexpr (MODE == PNEC_HIP_MODE_TARGET) {
    const double sgx = d[6] * gx + d[7] * gy + d[8] * gz;
    const double sgy = d[7] * gx + d[9] * gy + d[10] * gz;
    const double sgz = d[8] * gx + d[10] * gy + d[11] * gz;
    const double den = gx * sgx + gy * sgy + gz * sgz + reg;
    const double y = fast_rsqrt(fmax(den, kTinyDen));
    r = n * y;
    k = (r * y) * y;
  } else {
    constexpr bool kSym = (MODE == PNEC_HIP_MODE_SYM);
    const double ax = kSym ? f2x : f1x, ay = kSym ? f2y : f1y, az = kSym ? f2z : f1z;
    const double px = R[0] * ax + R[1] * ay + R[2] * az;
    const double py = R[3] * ax + R[4] * ay + R[5] * az;
    const double pz = R[6] * ax + R[7] * ay + R[8] * az;
    const double qx = U.t[1] * pz - U.t[2] * py;
    const double qy = U.t[2] * px - U.t[0] * pz;
    const double qz = U.t[0] * py - U.t[1] * px;
    constexpr int o = kSym ? 12 : 6;
    const double shx = d[o + 0] * qx + d[o + 1] * qy + d[o + 2] * qz;
    const double shy = d[o + 1] * qx + d[o + 3] * qy + d[o + 4] * qz;
    const double shz = d[o + 2] * qx + d[o + 4] * qy + d[o + 5] * qz;
    double den = qx * shx + qy * shy + qz * shz + reg;
    if constexpr (kSym) {
      const double sgx = d[6] * gx + d[7] * gy + d[8] * gz;
      const double sgy = d[7] * gx + d[9] * gy + d[10] * gz;
      const double sgz = d[8] * gx + d[10] * gy + d[11] * gz;
      den += gx * sgx + gy * sgy + gz * sgz;
    }
    const double y = fast_rsqrt(fmax(den, kTinyDen));
    r = n * y;
    k = (r * y) * y;
  }
}

__device__ __forceinline__ void accumulate(double r, const double (&J)[5], double (&acc)[kNumAcc]) {
  acc[0] = __builtin_fma(r, r, acc[0]);
#pragma unroll
  for (int a = 0; a < 5; ++a) acc[1 + a] = __builtin_fma(J[a], r, acc[1 + a]);
#pragma unroll
  for (int a = 0; a < 5; ++a)
#pragma unroll
    for (int b = a; b < 5; ++b) acc[6 + tri(a, b)] = __builtin_fma(J[a], J[b], acc[6 + tri(a, b)]);
}

// ------------------------------------------------------------------------------------------
// 5x5 SPD solve A y = b on the packed upper triangle (A(a,b) = P[tri(a,b)]); false if a pivot is
// not positive or the result is not finite (-> Ceres' LINEAR_SOLVER_FAILURE).
// The pivots are not tested one by one: a pivot <= 0 (or non-finite) makes its reciprocal square root
// NaN / inf (rsq(-x) = NaN, rsq(0) = inf and 0 * inf = NaN, rsq(inf) = 0 and inf * 0 = NaN), every
// later pivot and every z, y after it inherit that, and y[0] -- the last value of the back
// substitution -- depends on all of them: "y[0] is finite" is the whole test.
__device__ __forceinline__ bool chol_solve5(const double (&P)[15], const double (&b)[5],
                                            double (&y)[5]) {
  double L[15];  // L(i,j), i >= j, stored at tri(j,i)
  double inv[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    double dj = P[tri(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) dj = __builtin_fma(-L[tri(k, j)], L[tri(k, j)], dj);
    const double iv = fast_rsqrt(dj);
    inv[j] = iv;
    L[tri(j, j)] = dj * iv;
#pragma unroll
    for (int i = j + 1; i < 5; ++i) {
      double s = P[tri(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) s = __builtin_fma(-L[tri(k, i)], L[tri(k, j)], s);
      L[tri(j, i)] = s * iv;
    }
  }
  double z[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = __builtin_fma(-L[tri(k, i)], z[k], s);
    z[i] = s * inv[i];
  }
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    double s = z[i];
#pragma unroll
    for (int k = i + 1; k < 5; ++k) s = __builtin_fma(-L[tri(i, k)], y[k], s);
    y[i] = s * inv[i];
  }
  return finite_d(y[0]);
}

// The same 5x5 SPD system solved ACROSS LANES: lane i (i = 0..4) of every 16-lane row holds row i of A in the
// registers A[0..4] and the right-hand side b_i; on return b_i is x_i (lanes >= 5 hold garbage and are never read).
// Gauss-Jordan without pivoting (the pivots are those of the LDL' / Cholesky factorisation of an SPD matrix, so
// "every pivot > 0" is exactly "the Cholesky factorisation exists"): step j broadcasts row j -- fused into the
// multiply-adds (fmac_row) -- every other row eliminates its column j, and row j scales itself by 1 / pivot
// (multiplier 1 - 1/pivot on its own row), so no back substitution follows.  ~70 VALU slots against the ~125 of the
// replicated Cholesky + two triangular solves (the reciprocals dominate: 5 x 6).
// `damp` is added to the diagonal: lane i's damp to A(i,i).  A diagonal entry is only ever used as the pivot of its
// own step (until then it is updated like any other entry of its column), so the damping is added where the pivot
// is read -- one add per step instead of a per-lane select of "my diagonal register" up front.
// Returns the wave-uniform "all pivots positive".
__device__ __forceinline__ bool gj_solve5_rows(double (&A)[5], double damp, double &b, int li) {
  bool ok = true;
  auto step = [&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const double piv = bcast_row<j>(A[j] + damp);
    ok = ok && (piv > 0.0);
    const double inv = fast_rcp(piv);
    double f = A[j] * inv;
    f = (li == j) ? (1.0 - inv) : f;
    if constexpr (j < 1) fnmac_row<j>(A[1], A[1], f);
    if constexpr (j < 2) fnmac_row<j>(A[2], A[2], f);
    if constexpr (j < 3) fnmac_row<j>(A[3], A[3], f);
    if constexpr (j < 4) fnmac_row<j>(A[4], A[4], f);
    fnmac_row<j>(b, b, f);
  };
  step(std::integral_constant<int, 0>{});
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{});
  step(std::integral_constant<int, 4>{});
  return ok;
}

}  // namespace pnec_hip
