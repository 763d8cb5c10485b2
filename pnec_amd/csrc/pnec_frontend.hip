// pnec_frontend.hip -- the stages in front of the least-squares refinement:
//   PNEC::Eigensolver (use_ransac_ = false branch)   src/rel_pose_estimation/pnec.cc:273-278
//   PNEC::WeightedEigensolver                         src/rel_pose_estimation/pnec.cc:283-348
//   ComposeM / TranslationFromM / Weight              src/common/common.cc:127-136,157-181,183-208
//   fibonacci_sphere / obj_fun / scf                  src/optimization/scf.cc:43-72,109-148
//
// The eigensolver (opengv::relative_pose::eigensolver in the reference; opengv is not in the tree,
// so this is the published Kneip-Lynen algorithm: minimise the smallest eigenvalue of
// M(R) = sum (f1 x R f2)(f1 x R f2)' over the Cayley parameters of R) needs ONE pass over the
// payload: the 36 sums  G_kl[a][c] = sum_i w_i f2k f2l f1a f1c  determine
// M(R) = sum_kl [r_k]x G_kl [r_l]x' for every R (r_k = column k of R), so the damped-Newton
// iteration on the Cayley vector runs on 36 numbers parked in LDS.
//
// Work distribution (these stages are chains of one-value-per-pair FP64 work around short data-parallel
// passes, so the design question is how many lanes share one such chain):
//   * the eigenvalue minimiser (es_minimise_quad) runs on the four lanes of a quad: one evaluation gives
//     f, g at a point and the three finite-difference probes of the Hessian; the full Newton step is
//     tried as such a complete evaluation (one evaluation per iteration when it passes Armijo's test),
//     shorter steps four lengths at a time.  The smallest eigenpair of M comes from Rayleigh-quotient
//     iteration started at the neighbouring point's eigenvector or, without one, from the characteristic
//     polynomial (sym_eig3_min_rqi, sym_eig3_min_start), Jacobi sweeps only as the last resort;
//   * one minimisation per PAIR (plain / weighted eigensolver, RANSAC's eigensolver on the inliers) would
//     keep one quad of a wavefront busy, so those stages are split there: a wavefront per pair makes the 36
//     sums (sums36_kernel, the RANSAC kernel's inlier pass), es_batch_kernel minimises sixteen pairs per
//     wavefront, one per quad, and the rest of the stage starts from its result;
//   * weighted stage, pairs <= 512 correspondences: per correspondence n = f1 x R f2 and
//     B = f1hat R Sigma R' f1hat' + reg I are built ONCE per rotation and stay in registers; the 500
//     Fibonacci directions are pre-screened in packed single precision (table in constant memory,
//     scalar loads) and every candidate within 1e-4 of the best is re-evaluated in double precision;
//     SCF steps by Rayleigh-quotient iteration, stopped at the fixed point; once the rotation is final the
//     remaining rounds only redo the translation, and stop when a round leaves it bitwise unchanged;
//     pairs of up to 1024 / 2048 / 4096 correspondences run the same form on 2 / 4 / 8 wavefronts (sums exchanged
//     through LDS), chosen per pair in ragged batches; only larger ones stream the payload, 21 directions per pass;
//   * RANSAC: one wavefront per pair, one quad per hypothesis (16 per round), the quad's lanes sharing
//     the minimiser and splitting the correspondences when the hypothesis is scored; the sequential
//     consumption rule and the adaptive bound are wave-uniform scalar code.
// Reference quirks reproduced: C3 (weights from the initial pose in every iteration -- hence the
// rotation is final after the first converged eigensolver call), C4 (x1e-8), C5
// (E = sum A_i / t'B_i t), C6 (float division in fibonacci_sphere), C7 (ComposeM skips
// correspondence 0).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "pnec_device.hpp"
#include "pnec_front_shared.hpp"

namespace pnec_hip {

// -DPNEC_FRONT_DEBUG: event counters of the minimiser (diagnostics builds only; tools/build_front_variant.sh)
#ifdef PNEC_FRONT_DEBUG
__device__ unsigned long long g_dbg[24];
// wavefront-level event: counted once (x4, to match the per-quad print) by the first active lane
#define PNEC_DBG_WAVE(i) do { if ((int)threadIdx.x == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&g_dbg[i], 4ull); } while (0)
#define PNEC_DBG_COUNT(i) atomicAdd(&g_dbg[i], 1ull)
#else
#define PNEC_DBG_COUNT(i)
#define PNEC_DBG_WAVE(i)
#endif

// -DPNEC_WORK_COUNT: WORK counters of the front stages (a measurement build: tools/count_chain_work.py).  What a
// stage's roofline needs besides its duration is how much algorithmic work a launch held -- evaluations of the eigenvalue
// function, scored tiles, table builds, passes over the tables -- and that depends on the data (how many Newton
// iterations a hypothesis takes, where a model is dropped).  The counts are properties of the workload, not of the
// timing: they are taken once with this build, committed (profiles/chain_work_latest.json) and combined with the live
// stage times by bench.py.  The production build compiles every PNEC_WORK_ADD to nothing.
enum : int {
  kWkRansacEvals = 0,    // quad-evaluations (value + gradient at a point and its three Hessian probes, or four step lengths) of hypothesis minimisations
  kWkRansacHyps,         // hypotheses prepared (sample, 36 sums, jittered start)
  kWkRansacTiles,        // tiles of 64 correspondences scored against a model
  kWkRansacInlierCorr,   // correspondences of the final inlier pass
  kWkEsTailEvals,        // quad-evaluations of the eigensolver on the inliers (es_batch_kernel<translation>)
  kWkEsFirstEvals,       // ... of the plain / weighted stage's first minimisation (es_batch_kernel<none>)
  kWkWesEvals,           // ... of the weighted kernel's own later rounds
  kWkWesTableCorr,       // correspondences whose (n, B) table was built
  kWkWesCostCorr,        // correspondence terms of cost evaluations (current translation, candidate directions)
  kWkWesScfCorr,         // correspondence terms of SCF steps
  kWkWesBoundCorr,       // correspondence terms of the direction bound's matrix
  kWkSumsCorr,           // correspondences of the 36-sums passes (plain)
  kWkSumsWeightedCorr,   // ... weighted
  kWkCount = 16
};
#ifdef PNEC_WORK_COUNT
__device__ unsigned long long g_work[kWkCount];
#define PNEC_WORK_ADD(i, n) atomicAdd(&g_work[i], (unsigned long long)(n))
#else
#define PNEC_WORK_ADD(i, n) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// symmetric 3x3 eigen-decomposition, cyclic Jacobi; eigenvalues ascending, eigenvectors in the
// columns of V (row-major), largest-magnitude component of each made positive.
// Works on the 6 unique entries; the rotation angle comes from t = sgn(d) apq / (|d| + hypot(d, apq)),
// d = (aqq - app) / 2 -- the same t as 1 / (theta + sgn(theta) sqrt(theta^2 + 1)), theta = d / apq,
// without the division by a vanishing apq -- with v_rcp / v_rsq + Newton instead of IEEE divide/sqrt.
// WARM: V holds an orthonormal basis on entry (the eigenvectors of a nearby matrix): the sweeps
// run on V' A V, which is already almost diagonal, and converge in one or two instead of five.
template <bool WARM>
__device__ void sym_eig3_impl(const double (&A_in)[9], double (&w)[3], double (&V)[9]) {
  double A[9];
  if constexpr (WARM) {
    double B[9];  // B = A_in V
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c)
        B[3 * r + c] = A_in[3 * r] * V[c] + A_in[3 * r + 1] * V[3 + c] + A_in[3 * r + 2] * V[6 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = r; c < 3; ++c)
        A[3 * r + c] = A[3 * c + r] = V[r] * B[c] + V[3 + r] * B[3 + c] + V[6 + r] * B[6 + c];
  } else {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      A[i] = A_in[i];
      V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
  }
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-34 * dg || off == 0.0) break;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      constexpr int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2}, O[3] = {2, 1, 0};
      const int p = P[k], q = Q[k], o = O[k];  // rotate in the (p, q) plane; o = the third index
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double d = 0.5 * (A[3 * q + q] - A[3 * p + p]);
      const double hyp = fast_sqrt(__builtin_fma(d, d, apq * apq));
      double t = apq * fast_rcp(fabs(d) + hyp);
      t = (d < 0.0) ? -t : t;
      if (d == 0.0) t = 1.0;
      const double c = fast_rsqrt(__builtin_fma(t, t, 1.0)), sn = t * c;
      A[3 * p + p] = __builtin_fma(-t, apq, A[3 * p + p]);
      A[3 * q + q] = __builtin_fma(t, apq, A[3 * q + q]);
      A[3 * p + q] = A[3 * q + p] = 0.0;
      const double aop = A[3 * o + p], aoq = A[3 * o + q];
      A[3 * o + p] = A[3 * p + o] = c * aop - sn * aoq;
      A[3 * o + q] = A[3 * q + o] = sn * aop + c * aoq;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[3 * r + p], vrq = V[3 * r + q];
        V[3 * r + p] = c * vrp - sn * vrq;
        V[3 * r + q] = sn * vrp + c * vrq;
      }
    }
  }
  // ascending eigenvalues, their columns moved with them: three compare-and-swaps on (value, column) with selects (an
  // index permutation applied to V afterwards reads V by a run-time index, which puts the matrix in scratch memory)
  double d0 = A[0], d1 = A[4], d2 = A[8];
  double c0[3] = {V[0], V[3], V[6]}, c1[3] = {V[1], V[4], V[7]}, c2[3] = {V[2], V[5], V[8]};
  auto cswap = [](double &da, double &db, double (&ca)[3], double (&cb)[3]) {
    const bool sw = da > db;
    const double ta = sw ? db : da, tb = sw ? da : db;
    da = ta; db = tb;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const double xa = sw ? cb[r] : ca[r], xb = sw ? ca[r] : cb[r];
      ca[r] = xa; cb[r] = xb;
    }
  };
  cswap(d0, d1, c0, c1);
  cswap(d1, d2, c1, c2);
  cswap(d0, d1, c0, c1);
  auto put = [&](int c, double dv, const double (&col)[3]) {
    w[c] = dv;
    double big = col[0];
    if (fabs(col[1]) > fabs(big)) big = col[1];
    if (fabs(col[2]) > fabs(big)) big = col[2];
    const double sg = big < 0.0 ? -1.0 : 1.0;
    V[c] = sg * col[0];
    V[3 + c] = sg * col[1];
    V[6 + c] = sg * col[2];
  };
  put(0, d0, c0);
  put(1, d1, c1);
  put(2, d2, c2);
}
__device__ void sym_eig3(const double (&A)[9], double (&w)[3], double (&V)[9]) { sym_eig3_impl<false>(A, w, V); }

// Smallest eigenpair of a symmetric 3x3 from a nearby eigenvector (the previous point's, the
// previous SCF step's): Rayleigh-quotient iteration.  One step = the quotient mu = e'Me, the adjugate
// of M - mu I (six 2x2 minors: cross products of its rows; no division, so an exactly singular shift
// is harmless -- the adjugate is then the rank-one projector onto the eigenvector) applied to e, and a
// normalisation: ~50 FP64 instructions against ~300 per warm Jacobi sweep set; convergence is cubic,
// so a probe 1e-6 away needs one step and a line-search point two or three.  The iteration stops on the
// residual (below).  It can only be trusted to
// deliver the SMALLEST pair when it started near it, so the result is checked against the
// characteristic polynomial (lambda is the smallest root iff p'(lambda) >= 0 and trace - 3 lambda
// >= 0); on failure, or if it has not settled in 8 steps, the caller falls back to the Jacobi sweeps.
// Accuracy is the same kind as Jacobi's (eps * |M| in lambda, eps * |M| / gap in the vector).
__device__ __forceinline__ bool sym_eig3_min_rqi(const double (&M)[9], double (&e)[3], double &lambda) {
  const double m00 = M[0], m01 = M[1], m02 = M[2], m11 = M[4], m12 = M[5], m22 = M[8];
  const double tr = m00 + m11 + m22, atr = fabs(tr);
  double ex = e[0], ey = e[1], ez = e[2];
  double mx = 0.0, my = 0.0, mz = 0.0, mu = 0.0, res = 0.0;
  bool settled = false, lands = false;
#ifdef PNEC_RQI_CONFIRMING_STEP   // A/B: the form before round 4 (stop when a step moved the vector by < 1e-6)
  for (int step = 0; step < 8 && !settled; ++step) {
    PNEC_DBG_WAVE(8);
    mx = m00 * ex + m01 * ey + m02 * ez;
    my = m01 * ex + m11 * ey + m12 * ez;
    mz = m02 * ex + m12 * ey + m22 * ez;
    mu = ex * mx + ey * my + ez * mz;
    const double a00 = m00 - mu, a11 = m11 - mu, a22 = m22 - mu;
    const double c00 = a11 * a22 - m12 * m12, c01 = m02 * m12 - m01 * a22, c02 = m01 * m12 - m02 * a11;
    const double c11 = a00 * a22 - m02 * m02, c12 = m01 * m02 - m12 * a00, c22 = a00 * a11 - m01 * m01;
    double x = c00 * ex + c01 * ey + c02 * ez;
    double y = c01 * ex + c11 * ey + c12 * ez;
    double z = c02 * ex + c12 * ey + c22 * ez;
    const double n2 = x * x + y * y + z * z;
    if (!(n2 > 0.0) || !finite_d(n2)) return false;
    double inv = fast_rsqrt(n2);
    if (x * ex + y * ey + z * ez < 0.0) inv = -inv;
    x *= inv; y *= inv; z *= inv;
    const double moved = fmax(fabs(x - ex), fmax(fabs(y - ey), fabs(z - ez)));
    ex = x; ey = y; ez = z;
    settled = moved < 1e-6;
  }
  if (!settled) return false;
  mx = m00 * ex + m01 * ey + m02 * ez;
  my = m01 * ex + m11 * ey + m12 * ez;
  mz = m02 * ex + m12 * ey + m22 * ez;
  lambda = ex * mx + ey * my + ez * mz;
  res = fmax(fabs(mx - lambda * ex), fmax(fabs(my - lambda * ey), fabs(mz - lambda * ez)));
#else
  // Every trip starts with M e, the quotient and the residual |M e - mu e| of the vector it has -- what the end of the
  // iteration needs anyway.  It ends there when the residual is at rounding level, or when the residual BEFORE the
  // last step was below 1e-6 |tr|: convergence is cubic, that step landed at ~1e-18.  (Until round 4 the rule was
  // "a step moved the vector by < 1e-6", which spends a whole step on confirming what the residual says for a tenth
  // of its price: a probe 1e-6 away took two steps + the final products, now one step + two residuals; measured on
  // the RANSAC stage, as the wavefront executes them: 4.35 -> 2.9 steps per evaluation.)
  for (int step = 0; step < 9; ++step) {
    mx = m00 * ex + m01 * ey + m02 * ez;
    my = m01 * ex + m11 * ey + m12 * ez;
    mz = m02 * ex + m12 * ey + m22 * ez;
    mu = ex * mx + ey * my + ez * mz;
    res = fmax(fabs(mx - mu * ex), fmax(fabs(my - mu * ey), fabs(mz - mu * ez)));
    if (res <= 4e-15 * atr || lands) { settled = true; break; }
    if (step == 8) break;
    PNEC_DBG_WAVE(8);                  // eigen-iteration steps as the wavefront executes them
    lands = res <= 1e-6 * atr;
    const double a00 = m00 - mu, a11 = m11 - mu, a22 = m22 - mu;
    const double c00 = a11 * a22 - m12 * m12, c01 = m02 * m12 - m01 * a22, c02 = m01 * m12 - m02 * a11;
    const double c11 = a00 * a22 - m02 * m02, c12 = m01 * m02 - m12 * a00, c22 = a00 * a11 - m01 * m01;
    double x = c00 * ex + c01 * ey + c02 * ez;
    double y = c01 * ex + c11 * ey + c12 * ez;
    double z = c02 * ex + c12 * ey + c22 * ez;
    const double n2 = x * x + y * y + z * z;
    if (!(n2 > 0.0) || !finite_d(n2)) return false;
    double inv = fast_rsqrt(n2);
    if (x * ex + y * ey + z * ez < 0.0) inv = -inv;  // keep the orientation (adj is only defined up to sign)
    ex = x * inv; ey = y * inv; ez = z * inv;
  }
  if (!settled) return false;
  lambda = mu;
#endif
  // smallest root?  p(x) = x^3 - tr x^2 + c2 x - det:  p'(lambda) = (lambda - l2)(lambda - l3)
  const double c2 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12);
  const double dp = (3.0 * lambda - 2.0 * tr) * lambda + c2;
  const double tol = 1e-12 * tr * tr;
  if (!(dp >= -tol) || !(tr - 3.0 * lambda >= 0.0)) return false;
  // and it must BE an eigenpair: with a (nearly) double smallest eigenvalue the adjugate of M - mu I is a
  // difference of products that cancels to rounding noise, the iteration "settles" on a vector with 1e-7 of
  // the third eigenvector in it (device self-test, kind 1), and only the residual tells
  if (!(res <= 1e-13 * atr)) return false;
  e[0] = ex; e[1] = ey; e[2] = ez;
  return true;
}

// A start vector for sym_eig3_min_rqi when there is no neighbouring eigenvector (first evaluation of a
// minimisation) or the neighbour's led to another eigenpair: the smallest root of the characteristic
// polynomial q(x) = x^3 - tr x^2 + c2 x - det by Newton's method from x = 0 -- the matrices here are sums of
// n n' (positive semi-definite), so 0 is at or below the smallest eigenvalue and the iteration climbs to it
// monotonically; four steps leave it between 0 and lambda_1, much closer to lambda_1 than to lambda_2 unless
// the two nearly coincide -- then the adjugate of M - x I, which that shift makes (almost) the projector on the
// wanted eigenvector: its column with the largest diagonal entry, normalised.  ~95 instructions against
// ~900 for cold Jacobi sweeps; Rayleigh-quotient iteration polishes the pair and VERIFIES it (smallest root),
// so a start this recipe gets wrong (indefinite or degenerate input) still ends in the sweeps.
__device__ __forceinline__ void sym_eig3_min_start(const double (&M)[9], double (&e)[3]) {
  const double m00 = M[0], m01 = M[1], m02 = M[2], m11 = M[4], m12 = M[5], m22 = M[8];
  const double tr = m00 + m11 + m22;
  const double k00 = m11 * m22 - m12 * m12, k11 = m00 * m22 - m02 * m02, k22 = m00 * m11 - m01 * m01;
  const double c2 = k00 + k11 + k22;
  const double det = m00 * k00 - m01 * (m01 * m22 - m12 * m02) + m02 * (m01 * m12 - m11 * m02);
  double x = 0.0;
#pragma unroll
  for (int step = 0; step < 4; ++step) {
    const double q = __builtin_fma(__builtin_fma(x - tr, x, c2), x, -det);
    const double dq = __builtin_fma(__builtin_fma(3.0, x, -2.0 * tr), x, c2);
    x = __builtin_fma(-q, fast_rcp(dq), x);
  }
  const double a00 = m00 - x, a11 = m11 - x, a22 = m22 - x;
  const double c00 = a11 * a22 - m12 * m12, c01 = m02 * m12 - m01 * a22, c02 = m01 * m12 - m02 * a11;
  const double c11 = a00 * a22 - m02 * m02, c12 = m01 * m02 - m12 * a00, c22 = a00 * a11 - m01 * m01;
  const bool use0 = fabs(c00) >= fabs(c11) && fabs(c00) >= fabs(c22);
  const bool use1 = !use0 && fabs(c11) >= fabs(c22);
  const double x0 = use0 ? c00 : (use1 ? c01 : c02);
  const double x1 = use0 ? c01 : (use1 ? c11 : c12);
  const double x2 = use0 ? c02 : (use1 ? c12 : c22);
  const double inv = fast_rsqrt(x0 * x0 + x1 * x1 + x2 * x2);  // 0 -> inf -> NaN: the iteration rejects it
  e[0] = x0 * inv; e[1] = x1 * inv; e[2] = x2 * inv;
}

__device__ void cayley_to_rot(const double (&v)[3], double (&R)[9]) {
  const double x = v[0], y = v[1], z = v[2];
  const double s = fast_rcp(1.0 + x * x + y * y + z * z);
  R[0] = s * (1 + x * x - y * y - z * z); R[1] = s * 2 * (x * y - z); R[2] = s * 2 * (x * z + y);
  R[3] = s * 2 * (x * y + z); R[4] = s * (1 - x * x + y * y - z * z); R[5] = s * 2 * (y * z - x);
  R[6] = s * 2 * (x * z - y); R[7] = s * 2 * (y * z + x); R[8] = s * (1 - x * x - y * y + z * z);
}

__device__ __forceinline__ void cayley_to_rot_reduced(const double (&v)[3], double (&R)[9]) {
  const double x = v[0], y = v[1], z = v[2];
  R[0] = 1 + x * x - y * y - z * z; R[1] = 2 * (x * y - z); R[2] = 2 * (x * z + y);
  R[3] = 2 * (x * y + z); R[4] = 1 - x * x + y * y - z * z; R[5] = 2 * (y * z - x);
  R[6] = 2 * (x * z - y); R[7] = 2 * (y * z + x); R[8] = 1 - x * x - y * y + z * z;
}

__device__ void rot_to_cayley(const double (&R)[9], double (&v)[3]) {
  double A[9], B[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    A[i] = R[i] - (i % 4 == 0 ? 1.0 : 0.0);
    B[i] = R[i] + (i % 4 == 0 ? 1.0 : 0.0);
  }
  const double c00 = B[4] * B[8] - B[5] * B[7], c01 = B[5] * B[6] - B[3] * B[8], c02 = B[3] * B[7] - B[4] * B[6];
  const double det = B[0] * c00 + B[1] * c01 + B[2] * c02;
  double Bi[9];
  Bi[0] = c00 / det; Bi[1] = (B[2] * B[7] - B[1] * B[8]) / det; Bi[2] = (B[1] * B[5] - B[2] * B[4]) / det;
  Bi[3] = c01 / det; Bi[4] = (B[0] * B[8] - B[2] * B[6]) / det; Bi[5] = (B[2] * B[3] - B[0] * B[5]) / det;
  Bi[6] = c02 / det; Bi[7] = (B[1] * B[6] - B[0] * B[7]) / det; Bi[8] = (B[0] * B[4] - B[1] * B[3]) / det;
  // C = A B^-1; v = (-C(1,2), C(0,2), -C(0,1))
  v[0] = -(A[3] * Bi[2] + A[4] * Bi[5] + A[5] * Bi[8]);
  v[1] = A[0] * Bi[2] + A[1] * Bi[5] + A[2] * Bi[8];
  v[2] = -(A[0] * Bi[1] + A[1] * Bi[4] + A[2] * Bi[7]);
}

__device__ __forceinline__ void skew9(double x, double y, double z, double (&S)[9]) {
  S[0] = 0.0; S[1] = -z; S[2] = y;
  S[3] = z; S[4] = 0.0; S[5] = -x;
  S[6] = -y; S[7] = x; S[8] = 0.0;
}
__device__ __forceinline__ void mul33(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
// C = A B'
__device__ __forceinline__ void mul33t(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
// packed symmetric index of (a,c) in a 3x3: 00 01 02 11 12 22
__device__ __forceinline__ constexpr int s3(int a, int c) {
  return a <= c ? (a * 3 - a * (a - 1) / 2 + (c - a)) : (c * 3 - c * (c - 1) / 2 + (a - c));
}
__device__ __forceinline__ void cross3(const double (&a)[3], const double (&b)[3], double (&c)[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// M = sum_kl [r_k]x G_kl [r_l]x' from the 36 sums and the columns r_k of the rotation, as M = S + S' with
//   S = sum_k [r_k]x C_k,   C_k = G_kk [r_k / 2]x' + sum_{l > k} G_kl [r_l]x'
// (X_lk = X_kl' because the blocks are symmetric, and X_kk is symmetric).  Row i of G [b]x' is b x (row i of G), column j
// of [a]x C is a x (column j of C): every term is a cross product ACCUMULATED into its target -- two fused multiply-adds
// per component and nothing else: 6 blocks x 18 + 3 x 18 + 9 halvings + 6 additions = 177 instructions.  (Until round 6
// block by block, X_kl = [r_k]x G_kl [r_l]x' for k <= l and M += X_kl (+ X_kl') with the sums of the blocks as separate
// additions and a symmetrisation at the end: 310.)  The same matrix, symmetric by construction; last bits differ.
__device__ __forceinline__ void compose_m(const double (&Gr)[36], const double (&r)[3][3], double (&M)[9]) {
  double S[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) S[i] = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double Ck[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ck[i][0] = Ck[i][1] = Ck[i][2] = 0.0;
#pragma unroll
    for (int l = k; l < 3; ++l) {
      const double *Gp = Gr + 6 * s3(k, l);
      const double h = (l == k) ? 0.5 : 1.0;
      const double rx = h * r[l][0], ry = h * r[l][1], rz = h * r[l][2];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double g0 = Gp[s3(i, 0)], g1 = Gp[s3(i, 1)], g2 = Gp[s3(i, 2)];   // row i of the symmetric G_kl
        Ck[i][0] = __builtin_fma(ry, g2, __builtin_fma(-rz, g1, Ck[i][0]));
        Ck[i][1] = __builtin_fma(rz, g0, __builtin_fma(-rx, g2, Ck[i][1]));
        Ck[i][2] = __builtin_fma(rx, g1, __builtin_fma(-ry, g0, Ck[i][2]));
      }
    }
    const double kx = r[k][0], ky = r[k][1], kz = r[k][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const double c0 = Ck[0][j], c1 = Ck[1][j], c2 = Ck[2][j];              // column j of C_k
      S[j] = __builtin_fma(ky, c2, __builtin_fma(-kz, c1, S[j]));
      S[3 + j] = __builtin_fma(kz, c0, __builtin_fma(-kx, c2, S[3 + j]));
      S[6 + j] = __builtin_fma(kx, c1, __builtin_fma(-ky, c0, S[6 + j]));
    }
  }
  M[0] = S[0] + S[0]; M[4] = S[4] + S[4]; M[8] = S[8] + S[8];
  M[1] = M[3] = S[1] + S[3];
  M[2] = M[6] = S[2] + S[6];
  M[5] = M[7] = S[5] + S[7];
}

// lambda_min(M(R(v))) from the 36 sums, optionally its gradient w.r.t. the Cayley vector (e' dM e).
// M_out (row-major) is the composed matrix.  The 36 sums are read as G[i * GS]: GS = 1 for a
// table shared by the wavefront, GS = 64 for one table per lane interleaved in LDS (RANSAC).
//   M = sum_kl [r_k]x G_kl [r_l]x'   (G_kl symmetric 3x3, G_lk = G_kl)
//   composed by compose_m above.
// ew (optional): on entry the eigenvector of the smallest eigenvalue at a nearby point (`warm`), on
// exit the one at this point -- Rayleigh-quotient iteration from it instead of Jacobi sweeps from
// scratch (sym_eig3_min_rqi; falls back to the sweeps when it cannot vouch for the result).
#ifdef PNEC_ISA_MARKS
#define PNEC_FMARK(name) asm volatile("; PNEC_MARK " name)
#else
#define PNEC_FMARK(name)
#endif
// REDUCED: the rotation without its 1 / (1 + |v|^2) scale (opengv's math::cayley2rot_reduced, which is what its composeM
// is recalled to use [EXT]): M and lambda_min come out (1 + |v|^2)^2 times larger and the gradient is that function's
// (eigensolver scheme 2, pnec_es_schemes.inl).
template <int GS, bool REDUCED = false>
__device__ double es_value_grad(const double *G, const double (&v)[3], double *g, double *M_out,
                                double *ew = nullptr, bool warm = false) {
  // the 36 sums, read ONCE per evaluation, all loads in flight before the first use (read where they are used, the
  // composition of M waited for the table twelve times per evaluation -- two loads, s_waitcnt, seven instructions,
  // ... -- and the gradient read all of it a second time)
  double Gr[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) Gr[i] = G[i * GS];
  PNEC_FMARK("vg_rot");
  double R[9];
  if constexpr (REDUCED) cayley_to_rot_reduced(v, R);
  else cayley_to_rot(v, R);
  PNEC_FMARK("vg_M");
  double r[3][3];  // columns of R
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r[k][0] = R[k]; r[k][1] = R[3 + k]; r[k][2] = R[6 + k];
  }
  double M[9];
  compose_m(Gr, r, M);
  if (M_out) {
#pragma unroll
    for (int i = 0; i < 9; ++i) M_out[i] = M[i];
  }
  PNEC_FMARK("vg_eig");
  double lam = 0.0, e[3] = {0.0, 0.0, 1.0};
  bool have = false;
  if (ew) {
    // smallest eigenpair by Rayleigh-quotient iteration: from the neighbouring point's eigenvector (warm), and
    // from the characteristic polynomial's (sym_eig3_min_start) when there is none or it led elsewhere
    bool need_start = !warm;
    if (warm) { e[0] = ew[0]; e[1] = ew[1]; e[2] = ew[2]; }
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (need_start) { PNEC_DBG_WAVE(11); sym_eig3_min_start(M, e); }
      have = sym_eig3_min_rqi(M, e, lam);
      if (attempt == 0) {
        PNEC_DBG_COUNT(0);               // evaluations
        if (!have) PNEC_DBG_COUNT(1);    // ... whose first attempt failed
      }
      if (have || need_start) break;
      need_start = true;
    }
#ifdef PNEC_FRONT_DEBUG
    {
      const unsigned long long act = __builtin_amdgcn_ballot_w64(true), fb = __builtin_amdgcn_ballot_w64(!have);
      if ((int)threadIdx.x == __builtin_ctzll(act)) {
        atomicAdd(&g_dbg[6], 4ull);              // wavefront-level evaluations (x4 to match the /4 of the print)
        if (fb) atomicAdd(&g_dbg[7], 4ull);      // ... in which some lane went on to the Jacobi sweeps
      }
    }
#endif
  }
  if (!have) {
    double w[3], V[9];
    sym_eig3(M, w, V);
    lam = w[0];
    e[0] = V[0]; e[1] = V[3]; e[2] = V[6];
  }
  if (ew) { ew[0] = e[0]; ew[1] = e[1]; ew[2] = e[2]; }
  PNEC_FMARK("vg_grad");
  if (!g) return lam;
  // d lambda = e' dM e = 2 sum_k dr_k . q_k,  q_k = [e]x' (sum_l G_kl [e]x r_l) = (sum_l G_kl y_l) x e
  double y[3][3];
#pragma unroll
  for (int l = 0; l < 3; ++l) cross3(e, r[l], y[l]);
  double q[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double z[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const double *Gp = Gr + 6 * s3(k, l);
#pragma unroll
      for (int a = 0; a < 3; ++a)   // (one chain of fused multiply-adds per component: round 6)
        z[a] = __builtin_fma(Gp[s3(a, 0)], y[l][0], __builtin_fma(Gp[s3(a, 1)], y[l][1], __builtin_fma(Gp[s3(a, 2)], y[l][2], z[a])));
    }
    cross3(z, e, q[k]);
  }
  // d r_k / d v_j = (dN_j - 2 v_j R)[:, k] / s with N = (1 - |v|^2) I + 2 [v]x + 2 v v', s = 1 + |v|^2, and
  // dN_j = -2 v_j I + 2 [e_j]x + 2 (e_j v' + v e_j').  Contracted with Q[rr][k] = q_k[rr] term by term instead of forming
  // the three matrices (round 4: ~40 instructions instead of ~120 per evaluation):
  //   sum dN_j . Q = -2 v_j tr Q + 2 (Q[b][a] - Q[a][b]) + 2 sum_i v_i (Q[j][i] + Q[i][j]),   a = j + 1, b = j + 2 (mod 3)
  //   sum  R   . Q = the same number for every j
  const double inv_s = REDUCED ? 1.0 : fast_rcp(1.0 + v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double rq = 0.0;  // tr Q + sum R . Q   (REDUCED: d r_k / d v_j = dN_j[:, k], no R term and no scale: tr Q alone)
  if constexpr (!REDUCED) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) rq = __builtin_fma(R[3 * rr + k], q[k][rr], rq);
  }
  rq += q[0][0] + q[1][1] + q[2][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int a = (j + 1) % 3, b = (j + 2) % 3;
    double acc = q[a][b] - q[b][a];  // Q[b][a] - Q[a][b]
#pragma unroll
    for (int i = 0; i < 3; ++i) acc = __builtin_fma(v[i], q[i][j] + q[j][i], acc);
    acc = __builtin_fma(-v[j], rq, acc);
    g[j] = 4.0 * acc * inv_s;
  }
  return lam;
}

// M(R) from the 36 sums alone (es_value_grad's composition, without eigenpair and gradient): for the ordering key of
// es_queue_order below -- scheduling only, its bits enter no result.
__device__ __forceinline__ void sums_to_m(const double (&Gr)[36], const double (&R)[9], double (&M)[9]) {
  double r[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    r[k][0] = R[k]; r[k][1] = R[3 + k]; r[k][2] = R[6 + k];
  }
  compose_m(Gr, r, M);
}

// 3x3 Cholesky solve with reciprocal square roots (v_rsq_f64 + refinement) in place of the IEEE
// sqrt / divide sequences (~400 instructions per call otherwise)
__device__ bool solve3_spd(const double (&H)[9], const double (&b)[3], double (&x)[3]) {
  if (!(H[0] > 0.0)) return false;
  const double i0 = fast_rsqrt(H[0]);
  const double l10 = H[3] * i0, l20 = H[6] * i0;
  const double l11s = H[4] - l10 * l10;
  if (!(l11s > 0.0)) return false;
  const double i1 = fast_rsqrt(l11s);
  const double l21 = (H[7] - l20 * l10) * i1;
  const double l22s = H[8] - l20 * l20 - l21 * l21;
  if (!(l22s > 0.0)) return false;
  const double i2 = fast_rsqrt(l22s);
  const double z0 = b[0] * i0, z1 = (b[1] - l10 * z0) * i1, z2 = (b[2] - l20 * z0 - l21 * z1) * i2;
  x[2] = z2 * i2;
  x[1] = (z1 - l21 * x[2]) * i1;
  x[0] = (z0 - l10 * x[1] - l20 * x[2]) * i0;
  return true;
}

// The damped Newton direction of an iteration head: d = -(H + mu I)^-1 g.  mu = 0 when H is positive definite and the
// direction descends.  Otherwise mu = 2 |x|, x = a lower bound of H's smallest eigenvalue that is within a few percent
// of it unless eigenvalues nearly coincide: three Newton steps on the characteristic polynomial from the Gershgorin
// bound (from the left of the smallest root the iteration rises monotonically and never passes it).  H + mu I then has
// its smallest eigenvalue at about |lambda_min| -- neither nearly singular nor over-damped.  Should the factorisation
// still fail (rounding, a NaN), mu grows in decades, at most 40 tries in all; false = no direction (the caller ends the
// minimisation: the iterate stays).  The CPU checker's sequential form has the same rule.
//
// History (round 4).  The shift used to be SEARCHED: 0, 1e-6 tr, 1e-5 tr, ... until the factorisation went through.
// (i) Event counters showed 5.3 Cholesky solves per head as the wavefront executes them (an indefinite Hessian far from
// the minimum needs a shift of its own size, six or seven decades above the first, and the slowest of sixteen quads
// sets the count); walking over the shifts that leave a non-positive diagonal and trying four shifts at once on the
// quad's lanes brought that to 1.5, bit-identical.  (ii) The lengths of the minimisations have a heavy tail (7.4 Newton
// iterations on average, p99 21, 50 for the longest) and a round of 32 minimisations on sixteen quads is as long as its
// rare long one.  Traced on the checker, most long ones start on the flank of a saddle, where the searched shift -- the
// first decade above the threshold, i.e. anywhere between one and ten times |lambda_min| -- damps the step up to ten
// times too much and the iteration crawls with steps of g / mu for twenty or thirty iterations.  A finer search (x 2
// per try) lands close to the threshold too often: nearly singular solves, steps the Armijo search has to cut back,
// MORE evaluations on the device and worse parity.  The eigenvalue-based shift needs no search: counted on the checker
// over 2 400 RANSAC hypotheses, in evaluations as the device spends them, 9.96 -> 9.18 per minimisation, minimisations of
// 22 and more 2.6 % -> 0.7 %, a simulated round of 32 on sixteen quads 31 -> 27 evaluations long.
constexpr double kLevenbergGrowth = 10.0;
// A full, undamped Newton step shorter than this ends the minimisation: convergence is quadratic there, the point the
// step leads to is within ~C * 1e-12 of the minimiser (C = the ratio of third to second derivatives, 1..1e3 here), and
// the evaluation that would confirm it -- a whole trip of the quad, one in nine -- finds a step of 1e-12.  (Damped steps
// and cut-back steps say nothing of the kind and keep the old rule: 1e-12.)  Same constant in the CPU checker; counted
// there: 9.18 -> 8.61 evaluations per RANSAC minimisation, rotations within 1e-9 rad of the fully converged ones.
constexpr double kNewtonStepDone = 1e-6;
// ... and for the minimisation of a RANSAC HYPOTHESIS (ten correspondences; a model that is only scored against a
// threshold): the same.  Measured with 1e-4 (round 4): the last step of six minimisations in ten falls between 1e-6 and 1e-4,
// 8.68 -> 7.87 trips per minimisation on the checker and the stage 5 % faster on the device -- but checker and device,
// two floating-point realisations of one iteration, then part ways at the threshold's edge three times in 20 000 pairs
// (masks identical for 19 997; 1e-5: 19 999 and a gain within the noise) where with 1e-6 they do not once.  Not taken;
// the constant stays separate (same in the checker: ES_HYPOTHESIS_STEP_DONE).
constexpr double kHypothesisStepDone = 1e-6;
// Most Newton iterations of one minimisation: 50, and 25 for the minimisation of a RANSAC HYPOTHESIS (same constants in
// the CPU checker: ES_MAX_ITERATIONS, ES_HYPOTHESIS_MAX_ITERATIONS).  A minimisation converges in 4..15 iterations; the
// hypotheses that reach twenty and more are contaminated samples whose iterates crawl along the flank of a saddle
// (strongly negative curvature, a gradient with next to no component along it: steps of g / |lambda|, 1e-4 per iteration,
// for as long as they are allowed) or run off to the minimum at infinity of the Cayley chart (a rotation by 180 degrees).
// Next to never does one yield a round's best model, but one such minimisation sets the
// length of its round (53 trips where the round's others take ~27) and the 1 % of wavefronts that hold one are what a
// launch ends with.  (Until round 4: 50 everywhere.  The minimisations over a whole pair -- the eigensolver on the
// inliers, the weighted stage's -- keep 50: there the cap is met by ill-conditioned pairs bouncing at their noise floor,
// and where it cuts decides how far checker and device end apart.)
constexpr int kNewtonMaxIterations = 50;
constexpr int kHypothesisMaxIterations = 25;
// A hypothesis whose minimisation was cut off there YIELDS NO MODEL: it is consumed by the sequential rule with a count of
// zero, unscored (kModelCapped below; the checker: the same).  Cut off, checker and device stand at two different points
// of a long walk -- two floating-point realisations of the iteration drift apart over 25 steps that do not converge --
// and a model from either would be scored differently by the two; it happened to decide one pair in 20 000.  A count of
// zero is what such a model is worth (it never was a round's best on the checker), and it is the same on both sides.
constexpr int kModelCapped = 12;  // models[j][kModelCapped] != 0: hypothesis j's minimisation was cut off
__device__ __forceinline__ double hessian_floor(const double (&H)[9]) {
  const double m00 = H[0], m01 = H[1], m02 = H[2], m11 = H[4], m12 = H[5], m22 = H[8];
  const double trh = m00 + m11 + m22;
  const double c2 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12);
  const double det = m00 * (m11 * m22 - m12 * m12) - m01 * (m01 * m22 - m12 * m02) + m02 * (m01 * m12 - m11 * m02);
  double x = fmin(m00 - fabs(m01) - fabs(m02), fmin(m11 - fabs(m01) - fabs(m12), m22 - fabs(m02) - fabs(m12)));
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const double pq = ((x - trh) * x + c2) * x - det, dq = (3.0 * x - 2.0 * trh) * x + c2;
    if (dq > 0.0) x -= pq * fast_rcp(dq);   // (dq <= 0 cannot happen left of the smallest root; NaN ends up in mu and no try passes)
    else break;
  }
  return x;
}
__device__ __forceinline__ bool levenberg_direction(const double (&H)[9], const double (&g)[3], int role, double (&d)[3],
                                                    bool &damped) {
  (void)role;
  damped = false;
  const double tr = fabs(H[0]) + fabs(H[4]) + fabs(H[8]);
  const double mg[3] = {-g[0], -g[1], -g[2]};
  double mu = 0.0;
  for (int tries = 0; tries < 40; ++tries) {
    PNEC_DBG_WAVE(17);               // Cholesky solves as the wavefront executes them
    double Hm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hm[i] = H[i];
    Hm[0] += mu; Hm[4] += mu; Hm[8] += mu;
    if (solve3_spd(Hm, mg, d) && (d[0] * g[0] + d[1] * g[1] + d[2] * g[2]) < 0.0) return true;
    damped = true;
    mu = (tries == 0) ? fmax(2.0 * fmax(-hessian_floor(H), 0.0), 1e-6 * (tr + 1e-300)) : mu * kLevenbergGrowth;
  }
  return false;
}

// Damped Newton on the Cayley vector (opengv's eigensolver minimises lambda_min(M(R)) [EXT]; restated; the
// CPU checker under oracle/ holds the sequential form): gradient analytic (es_value_grad), Hessian by forward
// differences of the gradient (h = 1e-6), Levenberg shifts until it is positive definite, Armijo search over
// the step lengths 1, 1/2, 1/4, ...; returns the number of iterations taken (0 = already converged).
// Run by the four lanes of a quad (the callers give every quad its own problem -- RANSAC -- or all quads the
// same one -- NEC / weighted stage) which split what a single thread does in
// sequence: ONE evaluation yields f, g at a point (lane role 0) AND the three forward-difference probes
// of the Hessian (roles 1..3), or tries four step lengths of the Armijo search at once (alpha, alpha/2,
// alpha/4, alpha/8; the first that passes, in that order, wins -- the sequential rule).
//
// Written as a state machine with ONE evaluation per loop trip and a single call site of es_value_grad:
// the quads of a wavefront are at different points of their iterations (RANSAC: 16 hypotheses), and with
// one branch per kind of evaluation the wavefront executed every kind whenever any quad needed it -- 85 %
// of the trips ran the short-step search although only 11 % of the iterations reject the full step.  Now a
// trip costs one evaluation whatever the mix; a quad whose full step is rejected simply takes two more trips.
//   kInit / kReeval : f, g, H at v                      -> convergence test, Newton direction -> kTrial
//   kTrial          : f, g, H at v + d (the full step as a COMPLETE evaluation of the point it leads to)
//                     passes Armijo -> v += d, the iteration cost one evaluation;  fails -> kShort
//   kShort          : f at v + alpha d for four shorter lengths -> first that passes: v += alpha d, kReeval
// Same iterates as the sequential form up to the start vector of the eigen-iteration and the gradient rule
// of the full step (below); same trial sequence of step lengths (1, 1/2, 1/4, ..., at most 40 trials).
//
// TAG only separates instantiations: a non-inlined callee is compiled under the register budget of the
// kernels that call it, so a kernel that wants a different occupancy gets its own copy.
//
// e_out (optional): the eigenvector of the smallest eigenvalue of M at the returned v (unit length, sign
// arbitrary) -- every exit leaves the loop with the eigen-iteration's vector of exactly that point.
// active = false: this quad has no problem (its lanes only keep the wavefront's calls convergent): it is done at once.
template <int GS, int TAG = 0>
__device__ __noinline__ int es_minimise_quad(const double *G, double (&v)[3], double n_scale, double *e_out = nullptr,
                                             bool active = true, int *evals_out = nullptr,
                                             double step_done = kNewtonStepDone, int max_it = kNewtonMaxIterations) {
  enum : int { kInit = 0, kTrial, kShort, kReeval, kDone };
  const int role_of_lane = (int)(threadIdx.x & 3);
  const double h = 1e-6, inv_h = 1.0 / h;
  double eb[3] = {0.0, 0.0, 1.0};  // eigenvector of the smallest eigenvalue at the current point
  double f = 0.0, g[3] = {0.0, 0.0, 0.0}, H[9], d[3] = {0.0, 0.0, 0.0};
  double slope = 0.0, alpha = 1.0, trace_cur = 0.0;  // trace_cur: trace of M at the current point
  bool damped = false;                               // the direction d was made with a Levenberg shift
  int state = active ? kInit : kDone, it = 0, ls = 0, evals = 0;
  bool last_eval = false;
  while (state != kDone) {
    ++evals;  // diagnostics: evaluations of this quad's problem
    // ---- the point this lane evaluates in this trip (role: see es_minimise_queue)
    int role = role_of_lane;
    asm volatile("" : "+v"(role));
    double p[3] = {v[0], v[1], v[2]};
    double a_mine = 0.0;
    if (state == kShort) {
      a_mine = alpha * (role == 0 ? 1.0 : (role == 1 ? 0.5 : (role == 2 ? 0.25 : 0.125)));
#pragma unroll
      for (int k = 0; k < 3; ++k) p[k] = v[k] + a_mine * d[k];
    } else {
      if (state == kTrial) {
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = v[k] + d[k];
      }
      p[0] += (role == 1 ? h : 0.0);
      p[1] += (role == 2 ? h : 0.0);
      p[2] += (role == 3 ? h : 0.0);
    }
    double gp[3], Mp[9], ep[3] = {eb[0], eb[1], eb[2]};
    const double fp = es_value_grad<GS>(G, p, gp, Mp, ep, state != kInit);
    const double trace_p = Mp[0] + Mp[4] + Mp[8];  // Armijo's rounding-noise floor: lambda_min carries ~eps * trace(M)

    bool at_new_point = false;  // f, g, H, eb below describe v: run the head of the next iteration
    if (state == kShort) {
      PNEC_DBG_COUNT(5);               // short-step batches
      const int pass = (ls + role < 40 && fp <= f + 1e-4 * a_mine * slope + 4e-16 * trace_p) ? 1 : 0;
      const int p0 = quad_broadcast<0>(pass), p1 = quad_broadcast<1>(pass), p2 = quad_broadcast<2>(pass),
                p3 = quad_broadcast<3>(pass);
      if (p0 | p1 | p2 | p3) {
        alpha *= p0 ? 1.0 : (p1 ? 0.5 : (p2 ? 0.25 : 0.125));
        const double smax = alpha * fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] + alpha * d[k];
        ++it;
        // the loop ends here; with e_out the caller wants the eigenvector AT the new point: one more evaluation
        last_eval = smax < 1e-12 || it >= max_it;
        state = (last_eval && !e_out) ? kDone : kReeval;
      } else {
        alpha *= 0.0625;
        ls += 4;
        if (ls >= 40) state = kDone;  // no step length passes: the iterate stays
      }
    } else {
      // value, gradient and Hessian of the evaluated point from the quad
      const double fx = quad_broadcast<0>(fp), trace_x = quad_broadcast<0>(trace_p);
      double gx[3], Hx[9], ex[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        gx[r] = quad_broadcast<0>(gp[r]);
        Hx[3 * r + 0] = (quad_broadcast<1>(gp[r]) - gx[r]) * inv_h;
        Hx[3 * r + 1] = (quad_broadcast<2>(gp[r]) - gx[r]) * inv_h;
        Hx[3 * r + 2] = (quad_broadcast<3>(gp[r]) - gx[r]) * inv_h;
        ex[r] = quad_broadcast<0>(ep[r]);
      }
      Hx[1] = Hx[3] = 0.5 * (Hx[1] + Hx[3]);
      Hx[2] = Hx[6] = 0.5 * (Hx[2] + Hx[6]);
      Hx[5] = Hx[7] = 0.5 * (Hx[5] + Hx[7]);
      bool take = true;
      if (state == kTrial) {
        take = fx <= f + 1e-4 * slope + 4e-16 * trace_x;  // Armijo for the full step
        // M is composed from the 36 sums with cancellation (terms of the size of the sums add up to an
        // eigenvalue 1e-5 of it): its value carries ~50 eps trace(M) of noise, and within sqrt(noise / curvature)
        // ~ 1e-7 of the minimum the VALUE can no longer tell a good Newton step from a bad one.  The gradient
        // can (its noise moves the stationary point by ~1e-15): a full step that leaves the value unchanged
        // within that noise and shrinks the gradient is taken.  (The CPU oracle composes M from the
        // correspondences, sum of n n', without the cancellation, and needs no such rule: this is what makes
        // the device end where it does -- without it the device stopped up to 1.3e-7 rad short.)
        if (!take) {
          const double gmax_old = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
          const double gmax_new = fmax(fabs(gx[0]), fmax(fabs(gx[1]), fabs(gx[2])));
          take = (fx - f) <= 1e-13 * trace_x && gmax_new < gmax_old;
        }
        if (take) {
          const double smax = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
          for (int k = 0; k < 3; ++k) v[k] = v[k] + d[k];
          ++it;
          if (smax < (damped ? 1e-12 : step_done) || it >= max_it) state = kDone;
        } else {
          PNEC_DBG_COUNT(4);           // full step rejected
          state = kShort;
          alpha = 0.5;
          ls = 1;
        }
      }
      if (take) {
        f = fx;
        trace_cur = trace_x;
#pragma unroll
        for (int i = 0; i < 3; ++i) { g[i] = gx[i]; eb[i] = ex[i]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) H[i] = Hx[i];
        if (last_eval) state = kDone;
        at_new_point = state != kDone;
      }
    }
    if (at_new_point) {
      // ---- head of a Newton iteration at v: converged?  else the damped Newton direction
      const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
#ifdef PNEC_FRONT_TRACE_ITER
      if (threadIdx.x == 0 && blockIdx.x == 0)
        printf("it %d f %.17g gmax %.3e trace %.3e tol %.3e floor %.3e H %.3e %.3e %.3e | %.3e %.3e %.3e\n", it, f, gmax, trace_cur,
               1e-14 * (1.0 + fabs(f)) * n_scale, 8.9e-16 * trace_cur, H[0], H[4], H[8], H[1], H[2], H[5]);
#endif
      // converged: the gradient tolerance of the sequential form -- or the gradient's own noise floor when that
      // is the larger.  g = e' dM e is composed from sums of the size of trace(M); traced on the device, what is
      // left of it bounces at 100..300 eps trace(M) once the iteration has arrived.  For ordinary problems that
      // is far below the tolerance (RANSAC samples: trace 0.02, floor 2.5e-15 against 1e-13; weighted pairs: trace
      // 1..10, floor <= 1e-12 against 4.5e-12).  It decides when one correspondence's weight dwarfs the rest
      // (trace 4e4 from a nearly singular covariance): there the gradient bounced between 7e-11 and 4e-9 with
      // steps of 1e-11 -- too large for the step-size stop -- until the cap of 50, and again in each of the nine
      // rounds; one such pair in 20 000 set the duration of two launches.
      if (gmax <= fmax(1e-14 * (1.0 + fabs(f)) * n_scale, 1.1e-13 * trace_cur)) {
        state = kDone;
      } else {
        PNEC_DBG_COUNT(2);             // Newton iterations (x4 lanes)
        PNEC_DBG_WAVE(10);             // iteration heads as the wavefront executes them
        const bool ok = levenberg_direction(H, g, role, d, damped);
        if (ok) {
          slope = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
          state = kTrial;
        } else {
          state = kDone;
        }
      }
    }
  }
  if (e_out) { e_out[0] = eb[0]; e_out[1] = eb[1]; e_out[2] = eb[2]; }
  if (evals_out) *evals_out = evals;
#ifdef PNEC_WORK_COUNT
  // TAG 1: a RANSAC hypothesis (one problem per quad); TAG 2: es_batch_kernel (one pair per quad; which epilogue: the
  // caller moves the count); TAG 0: the weighted kernel's in-kernel rounds (every quad repeats the pair's problem: once)
  if (TAG == 1 && role_of_lane == 0 && active) PNEC_WORK_ADD(kWkRansacEvals, evals);
  if (TAG == 0 && threadIdx.x == 0 && active) PNEC_WORK_ADD(kWkWesEvals, evals);
#endif
  return it;
}

// ------------------------------------------------------------------------------------------
constexpr int kFibStride = 9;  // t (3) | txx tyy tzz | 2 txy, 2 txz, 2 tyz
// the 500 Fibonacci directions with their products, in CONSTANT memory: the search loop's index is wave-
// uniform, and loads from the constant address space are scalar loads (s_load: no VGPRs, no vector-memory
// latency in the loop -- through a plain global pointer the compiler issued five vector loads per
// direction and waited for them, which was half of the stage's wall time)
__constant__ double c_fib[500 * kFibStride];
__constant__ float c_fib32[500 * kFibStride];  // the same table rounded to single precision (pre-screen of the rare pair)
// the directions' products txx tyy tzz | 2 txy 2 txz 2 tyz, one PLANE per product (the bound test of the weighted stage
// reads them one direction per lane: 64 consecutive doubles per load instead of 64 lines); entries 500..511 are zero
__device__ double c_fib_prod[6][512];
// this lane's directions 64 (J0 + j) + lane, j = 0..3: their six products each from the planes above, in the scalar-base form of
// global_load (see pnec_device.hpp load_planes_saddr for why): 24 loads and their wait in ONE statement
#define PNEC_FIB_SET_(J, OFF)                                                                              \
  PNEC_LD_(f##J##_0, b0, OFF) PNEC_LD_(f##J##_1, b1, OFF) PNEC_LD_(f##J##_2, b2, OFF) PNEC_LD_(f##J##_3, b3, OFF) \
  PNEC_LD_(f##J##_4, b4, OFF) PNEC_LD_(f##J##_5, b5, OFF)
#define PNEC_FIB_OUT_(J)                                                                                   \
  [f##J##_0] "+v"(fp[J][0]), [f##J##_1] "+v"(fp[J][1]), [f##J##_2] "+v"(fp[J][2]), [f##J##_3] "+v"(fp[J][3]),  \
  [f##J##_4] "+v"(fp[J][4]), [f##J##_5] "+v"(fp[J][5])
template <int J0>
__device__ __forceinline__ void fib_prod_load4(double (&fp)[4][6], const char *planes, unsigned voff) {
  static_assert(J0 == 0 || J0 == 4, "the grid's two halves");
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 6; ++k) fp[j][k] = 0.0;
  constexpr size_t pb = 512 * sizeof(double);
  if constexpr (J0 == 0)
    asm volatile("s_nop 4\n\t" PNEC_FIB_SET_(0, "0") PNEC_FIB_SET_(1, "512") PNEC_FIB_SET_(2, "1024") PNEC_FIB_SET_(3, "1536")
                 "s_waitcnt vmcnt(0)"
                 : PNEC_FIB_OUT_(0), PNEC_FIB_OUT_(1), PNEC_FIB_OUT_(2), PNEC_FIB_OUT_(3)
                 : [lo] "v"(voff), [b0] "s"(planes), [b1] "s"(planes + pb), [b2] "s"(planes + 2 * pb), [b3] "s"(planes + 3 * pb),
                   [b4] "s"(planes + 4 * pb), [b5] "s"(planes + 5 * pb)
                 : "memory");
  else
    asm volatile("s_nop 4\n\t" PNEC_FIB_SET_(0, "2048") PNEC_FIB_SET_(1, "2560") PNEC_FIB_SET_(2, "3072") PNEC_FIB_SET_(3, "3584")
                 "s_waitcnt vmcnt(0)"
                 : PNEC_FIB_OUT_(0), PNEC_FIB_OUT_(1), PNEC_FIB_OUT_(2), PNEC_FIB_OUT_(3)
                 : [lo] "v"(voff), [b0] "s"(planes), [b1] "s"(planes + pb), [b2] "s"(planes + 2 * pb), [b3] "s"(planes + 3 * pb),
                   [b4] "s"(planes + 4 * pb), [b5] "s"(planes + 5 * pb)
                 : "memory");
}
typedef float v2f __attribute__((ext_vector_type(2)));
// 1/x to ~1e-14 (seed + one Newton step): for the direction search, whose winner is re-evaluated exactly
__device__ __forceinline__ double rcp_search(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
}
struct FrontArgs {
  const double *data;           // SoA payload (12 planes for the weighted stage, >= 6 for NEC)
  const int64_t *block_offset;
  const int32_t *count;
  const double *init_q;  // [n_pairs,4] xyzw
  const double *init_t;  // [n_pairs,3] (weighted stage)
  double *out_q;         // [n_pairs,4]
  double *out_t;         // [n_pairs,3]
  int32_t *out_iterations;  // [n_pairs] Newton iterations of the (first) eigensolver call, or null
  unsigned long long *trace;  // null, or [n_pairs, 8] clocks per phase of the weighted kernel (PNEC_HIP_TRACE_FRONT)
  double reg;
  int weighted_iterations;
  // weighted stage: ALL its eigenvalue minimisations ran before it, chained (es_batch_kernel / es_batch_alt_kernel) --
  // the first call's iteration count [n_pairs], round r's minimiser of pair p at pre_v_rounds[3 (r n_pairs + p)], and
  // how many rounds of the pair have one (after those the rotation is final) at pre_n_es[p]
  const int32_t *pre_its;
  const int32_t *order;    // [n_pairs] the pair block b takes (weighted_order_place), or null: pair b
  const double *pre_v_rounds;
  const int32_t *pre_n_es;
  int64_t n_pairs;
};
// phase clocks of the weighted kernel (diagnostics): s_memtime differences accumulated per phase
enum : int { kPhSums = 0, kPhNewton, kPhTables, kPhSearch, kPhCost, kPhScf, kPhTotal, kPhCount = 12 };
#define PNEC_PHASE_BEGIN() unsigned long long ph_t0_ = a.trace ? __builtin_amdgcn_s_memtime() : 0ull
#define PNEC_PHASE_END(ph)                                                  \
  do {                                                                      \
    if (a.trace) {                                                          \
      const unsigned long long ph_t1_ = __builtin_amdgcn_s_memtime();       \
      ph_clk[ph] += ph_t1_ - ph_t0_;                                        \
      ph_t0_ = ph_t1_;                                                      \
    }                                                                       \
  } while (0)

// (the largest-diagonal cases are written out per i: indexing R by a run-time i put the CALLER's rotation -- nine doubles
// that live across the whole kernel -- into scratch memory, loaded and stored once per outer iteration)
template <int I>
__device__ __forceinline__ void quat_from_rot_case(const double (&R)[9], double (&q)[4]) {
  constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
  double t = sqrt(R[4 * I] - R[4 * J] - R[4 * K] + 1.0);
  q[I] = 0.5 * t;
  t = 0.5 / t;
  q[3] = (R[3 * K + J] - R[3 * J + K]) * t;
  q[J] = (R[3 * J + I] + R[3 * I + J]) * t;
  q[K] = (R[3 * K + I] + R[3 * I + K]) * t;
}
__device__ __forceinline__ void quat_from_rot_dev(const double (&R)[9], double (&q)[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0.0) {
    double t = sqrt(tr + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
    if (i == 0) quat_from_rot_case<0>(R, q);
    else if (i == 1) quat_from_rot_case<1>(R, q);
    else quat_from_rot_case<2>(R, q);
  }
}

// 36 sums G_kl[a][c] over the pair; WEIGHTED multiplies by Weight(init pose) * 1e-8 (C3, C4)
template <bool WEIGHTED>
__device__ void pass_sums36(const double *base, int n, int stride, const double (&R0)[9],
                            const double (&t0)[3], double reg, int lane, double *G /* LDS, 36 */) {
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  for (int idx = lane; idx < stride; idx += kWave) {
    const double f1[3] = {base[idx], base[(int64_t)stride + idx], base[(int64_t)2 * stride + idx]};
    const double f2[3] = {base[(int64_t)3 * stride + idx], base[(int64_t)4 * stride + idx],
                          base[(int64_t)5 * stride + idx]};
    double w = 1.0;
    if constexpr (WEIGHTED) {
      const double mx = t0[1] * f1[2] - t0[2] * f1[1], my = t0[2] * f1[0] - t0[0] * f1[2],
                   mz = t0[0] * f1[1] - t0[1] * f1[0];
      const double gx = R0[0] * mx + R0[3] * my + R0[6] * mz, gy = R0[1] * mx + R0[4] * my + R0[7] * mz,
                   gz = R0[2] * mx + R0[5] * my + R0[8] * mz;
      const double sxx = base[(int64_t)6 * stride + idx], sxy = base[(int64_t)7 * stride + idx],
                   sxz = base[(int64_t)8 * stride + idx], syy = base[(int64_t)9 * stride + idx],
                   syz = base[(int64_t)10 * stride + idx], szz = base[(int64_t)11 * stride + idx];
      const double q = gx * (sxx * gx + sxy * gy + sxz * gz) + gy * (sxy * gx + syy * gy + syz * gz) +
                       gz * (sxz * gx + syz * gy + szz * gz);
      w = (idx < n) ? (1.0 / (q + reg)) * 1.0e-8 : 0.0;
    }
    double p[6], qq[6];
    p[0] = w * f2[0] * f2[0]; p[1] = w * f2[0] * f2[1]; p[2] = w * f2[0] * f2[2];
    p[3] = w * f2[1] * f2[1]; p[4] = w * f2[1] * f2[2]; p[5] = w * f2[2] * f2[2];
    qq[0] = f1[0] * f1[0]; qq[1] = f1[0] * f1[1]; qq[2] = f1[0] * f1[2];
    qq[3] = f1[1] * f1[1]; qq[4] = f1[1] * f1[2]; qq[5] = f1[2] * f1[2];
#pragma unroll
    for (int kl = 0; kl < 6; ++kl)
#pragma unroll
      for (int ac = 0; ac < 6; ++ac) acc[6 * kl + ac] = __builtin_fma(p[kl], qq[ac], acc[6 * kl + ac]);
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) {
    const double s = wave_allreduce_sum(acc[i]);
    if (lane == 0) G[i] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
}

// ---- The eigenvalue minimisation of a pair is ONE chain of dependent evaluations that keeps one quad busy
// (es_minimise_quad).  Run inside a kernel that owns a wavefront per pair, fifteen of the sixteen quads
// repeat it redundantly -- 18 % of the weighted stage, 6 % of RANSAC, 90 % of the plain eigensolver went
// there.  So the stages are split at that point: a wavefront per pair produces the pair's 36 sums
// (sums36_kernel, or the RANSAC kernel's inlier pass), es_batch_kernel then minimises SIXTEEN pairs per
// wavefront, one per quad, and the rest of the stage carries on from its result.  Per pair the arithmetic is
// exactly what it was (same sums, same minimiser on the same lanes of a quad): the results do not change.
constexpr int kStBuckets = 8;   // lists of the RANSAC stage's second launch (see ransac2_eigensolver_kernel, PHASE)
static_assert(4 + kStBuckets <= kFrontCounterInts && 58 + kStBuckets / 2 <= kFrontDoublesPerPair, "front scratch layout");
struct FrontScratch {
  double *G;        // [P,36] the sums of every pair
  double *v0;       // [P,3]  Cayley vector the minimisation starts from
  double *n_scale;  // [P]    number of correspondences in the sums (>= 1): scales the gradient tolerance
  double *v;        // [P,3]  the minimiser
  int32_t *its;     // [P]    Newton iterations taken
  int32_t *first;   // [P]    correspondence ComposeM leaves out (C7), -1: none
  double *v_rounds; // [kEsMaxRounds,P,3] schemes 1, 2: the weighted stage's minimiser of every round
  int32_t *n_es;    // [P]    ... and how many rounds have one
  // the RANSAC stage's two launches (ransac2_eigensolver_kernel PHASE 1 / 2): the rule's state per pair lives in the
  // v_rounds region (free until the weighted stage), the list of pairs that go on and its length behind the ints
  double *st_k, *st_it, *st_best, *st_model;
  int32_t *st_list, *st_count;   // [kStBuckets, P], [kStBuckets]
  // the weighted stage's launch order (same ints, free by then): wo_order[b] = the pair block b of the weighted kernel
  // takes, wo_count[0 / 1] = pairs placed from the front / from the back so far
  int32_t *wo_order, *wo_count;
};
// per pair kFrontDoublesPerPair = 36 + 3 + 1 + 3 + 3 kEsMaxRounds doubles and kFrontIntsPerPair ints, and kFrontCounterInts
// ints of counters behind those (pnec_front_shared.hpp; pnec_capi.hip allocates them -- until round 5 the counters sat
// in a fifth per-pair region, i.e. past the end of a one-pair batch's five ints)
FrontScratch front_scratch(double *d, int32_t *i, int64_t P) {
  FrontScratch f;
  f.G = d;
  f.v0 = d + 36 * P;
  f.n_scale = d + 39 * P;
  f.v = d + 40 * P;
  f.v_rounds = d + 43 * P;
  f.its = i;
  f.first = i + P;
  f.n_es = i + 2 * P;
  f.st_k = d + 43 * P;
  f.st_it = d + 44 * P;
  f.st_best = d + 45 * P;
  f.st_model = d + 46 * P;   // .. 58 P (of the 45 P the v_rounds region has)
  // kStBuckets lists of up to P pair indices each (a pair is listed in the bucket of the rounds its rule still asks for),
  // 8 P ints = 4 P doubles behind the rule's state: 58 P .. 62 P of the 88 P doubles
  f.st_list = reinterpret_cast<int32_t *>(d + 58 * P);
  int32_t *counters = i + (int64_t)kFrontIntsPerPair * P;
  f.st_count = counters + 4; // (kStBuckets ints: counters[4 .. 12))
  f.wo_order = i + 3 * P;
  f.wo_count = counters + 2; // (two ints)
  return f;
}

// 36 sums of every pair from its start rotation (WEIGHTED: and translation -- the weights of C3 / C4)
template <bool WEIGHTED>
__global__ __launch_bounds__(kWave) void sums36_kernel(const FrontArgs a, const FrontScratch out) {
  const int64_t pair = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = a.count[pair];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const double *base = a.data + a.block_offset[pair];
  __shared__ double G[36];
  double q0[4] = {a.init_q[4 * pair], a.init_q[4 * pair + 1], a.init_q[4 * pair + 2], a.init_q[4 * pair + 3]};
  {
    const double qn = 1.0 / sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    for (int k = 0; k < 4; ++k) q0[k] *= qn;
  }
  double R0[9];
  rot_from_quat(q0, R0);
  double t0[3] = {0.0, 0.0, 1.0};
  if constexpr (WEIGHTED) {
    t0[0] = a.init_t[3 * pair]; t0[1] = a.init_t[3 * pair + 1]; t0[2] = a.init_t[3 * pair + 2];
  }
  pass_sums36<WEIGHTED>(base, n, stride, R0, t0, WEIGHTED ? a.reg : 0.0, lane, G);
  if (lane == 0) PNEC_WORK_ADD(WEIGHTED ? kWkSumsWeightedCorr : kWkSumsCorr, n);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (lane < 36) out.G[36 * pair + lane] = G[lane];
  if (lane == 0) {
    double v[3];
    rot_to_cayley(R0, v);
    out.v0[3 * pair] = v[0]; out.v0[3 * pair + 1] = v[1]; out.v0[3 * pair + 2] = v[2];
    out.n_scale[pair] = (double)(n > 0 ? n : 1);
    out.first[pair] = n > 0 ? 0 : -1;  // ComposeM starts at i = 1 (C7)
    if (WEIGHTED && pair == 0) out.wo_count[0] = out.wo_count[1] = 0;  // (es_batch_kernel, the next launch, counts)
  }
}

struct EsBatchArgs {
  FrontScratch s;
  int64_t n_pairs;
  // translation epilogue (PNEC::Eigensolver's tail: ComposeM without correspondence `first`, TranslationFromM)
  const double *data;
  const int64_t *block_offset;
  const int32_t *count;
  double *out_q, *out_t;
  int32_t *out_iterations;  // Newton iterations, or null
};
constexpr int kEpiNone = 0, kEpiTranslation = 1;
static unsigned es_batch_blocks(int64_t n_pairs) { return (unsigned)((n_pairs + 15) / 16); }

// PNEC::Eigensolver's tail for the pair of a quad (es_batch_kernel<kEpiTranslation>): the rotation as a quaternion and the
// translation = eigenvector of the smallest eigenvalue of M without the correspondence ComposeM skips (C7)
__device__ __forceinline__ void es_batch_translation_epilogue(const EsBatchArgs &a, const double *Gq, const double (&v)[3],
                                                              int64_t pair, bool mine, int it, int lane) {
  double M[9], R[9];
  es_value_grad<1>(Gq, v, nullptr, M);
  cayley_to_rot(v, R);
  const int f = a.s.first[pair];
  if (f >= 0) {
    const int n = a.count[pair];
    const int stride = (n + kWave - 1) & ~(kWave - 1);
    const double *base = a.data + a.block_offset[pair];
    const double f1[3] = {base[f], base[(int64_t)stride + f], base[(int64_t)2 * stride + f]};
    const double f2[3] = {base[(int64_t)3 * stride + f], base[(int64_t)4 * stride + f], base[(int64_t)5 * stride + f]};
    const double u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                         R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
    const double nn[3] = {f1[1] * u[2] - f1[2] * u[1], f1[2] * u[0] - f1[0] * u[2], f1[0] * u[1] - f1[1] * u[0]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[3 * r + c] -= nn[r] * nn[c];
  }
  double w[3], V[9];
  sym_eig3(M, w, V);
  if (mine && (lane & 3) == 0) {
    double qo[4];
    quat_from_rot_dev(R, qo);
    const double qn = 1.0 / sqrt(qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3]);
    for (int k = 0; k < 4; ++k) a.out_q[4 * pair + k] = qo[k] * qn;
    const double tn = 1.0 / sqrt(V[0] * V[0] + V[3] * V[3] + V[6] * V[6]);
    a.out_t[3 * pair + 0] = V[0] * tn;
    a.out_t[3 * pair + 1] = V[3] * tn;
    a.out_t[3 * pair + 2] = V[6] * tn;
    if (a.out_iterations) a.out_iterations[pair] = it;
  }
}

// The weighted kernel's launch order.  A launch ends with its slowest pair, and the slow pairs of that kernel are the ones
// whose translation is barely observable: the cost is nearly flat along a second direction, the bound prunes little of the
// Fibonacci grid (the full-grid search: 6-12 times a median pair's clocks) and the SCF wanders through all its rounds.
// Dispatched in index order, 1-2 % of such pairs left the last THIRD of a 20 000-pair launch to a few dozen wavefronts
// (slots busy on average: 1 090 of 2 048; round 5, tools/analyse_weighted_trace.py).  How flat the cost is shows BEFORE that
// kernel runs: smallest / second eigenvalue of the weighted M at the first minimiser (the top 5 % by that ratio held 298 of
// the 302 pairs that took more than three medians).  So the pairs above a fixed ratio are placed from the front of the
// order and the others from the back -- one atomic per pair, no sort; the order inside the two classes is whatever the
// atomics give (it decides WHEN a pair runs, never what comes out: every pair writes its own records).
constexpr double kWeightedLongRatio = 1e-3;
// called by the whole wavefront; `place`: this lane holds a pair (and its eigenvalues w).  One atomic per class and
// WAVEFRONT (one per pair, all on two addresses, cost the launch 150 us per 20 000 pairs)
__device__ __forceinline__ void weighted_order_place(const FrontScratch &s, int64_t n_pairs, bool place, int64_t pair,
                                                     const double (&w)[3]) {
  const int lane = (int)threadIdx.x & (kWave - 1);
  const bool long_pair = place && !(w[0] < kWeightedLongRatio * w[1]);  // (a NaN counts as long: nothing is known about it)
  const bool short_pair = place && !long_pair;
  const unsigned long long ml = __builtin_amdgcn_ballot_w64(long_pair), ms = __builtin_amdgcn_ballot_w64(short_pair);
  int base_l = 0, base_s = 0;
  if (lane == 0) {
    if (ml != 0ull) base_l = atomicAdd(&s.wo_count[0], __builtin_popcountll(ml));
    if (ms != 0ull) base_s = atomicAdd(&s.wo_count[1], __builtin_popcountll(ms));
  }
  base_l = __builtin_amdgcn_readfirstlane(base_l);
  base_s = __builtin_amdgcn_readfirstlane(base_s);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (long_pair) s.wo_order[base_l + __builtin_popcountll(ml & below)] = (int32_t)pair;
  if (short_pair) s.wo_order[n_pairs - 1 - base_s - __builtin_popcountll(ms & below)] = (int32_t)pair;
}

// sixteen pairs per wavefront, one per quad: minimise lambda_min(M(R)) from v0; kEpiTranslation: then the
// rotation as a quaternion and the translation = eigenvector of the smallest eigenvalue of M without the
// correspondence ComposeM skips
// kEpiNone (the weighted stage): ALL of the stage's minimisations, chained -- the weights never change (C3), so a later
// round minimises the same function from the previous round's result, and a call that ended for any other reason than
// the iteration cap leaves nothing to do: normally ONE call per pair; a pair at the cap goes on (the other quads idle
// through it), `rounds` calls at most.  Round r's minimiser at v_rounds[r], the number of rounds that have one at n_es.
template <int EPI>
__global__ __launch_bounds__(kWave, 2) void es_batch_kernel(const EsBatchArgs a, int rounds) {
  const int lane = threadIdx.x;
  const int quad = lane >> 2;
  __shared__ double Gs[16][36];
  const int64_t first_pair = 16 * (int64_t)blockIdx.x;
  for (int i = lane; i < 16 * 36; i += kWave) {
    int64_t p = first_pair + i / 36;
    p = p < a.n_pairs ? p : a.n_pairs - 1;  // the last wavefront's spare quads repeat the last pair (and write nothing)
    Gs[i / 36][i % 36] = a.s.G[36 * p + i % 36];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool mine = first_pair + quad < a.n_pairs;
  const int64_t pair = mine ? first_pair + quad : a.n_pairs - 1;
  double v[3] = {a.s.v0[3 * pair], a.s.v0[3 * pair + 1], a.s.v0[3 * pair + 2]};
  int it = 0, n_es = 0;
  bool going = mine;
  for (int r = 0; r < (EPI == kEpiNone ? rounds : 1); ++r) {
    if (r > 0 && __builtin_amdgcn_ballot_w64(going) == 0ull) break;
    const double v_in[3] = {v[0], v[1], v[2]};
#ifdef PNEC_WORK_COUNT
    int evals_here = 0;
    const int it_r = es_minimise_quad<1, 3>(Gs[quad], v, a.s.n_scale[pair], nullptr, true, &evals_here);   // (TAG 3: counted here)
    if (going && (lane & 3) == 0) PNEC_WORK_ADD(EPI == kEpiNone ? kWkEsFirstEvals : kWkEsTailEvals, evals_here);
#else
    const int it_r = es_minimise_quad<1, 2>(Gs[quad], v, a.s.n_scale[pair]);
#endif
    if (r == 0) it = it_r;
    if (!going) {  // a quad that only kept the others company
      v[0] = v_in[0]; v[1] = v_in[1]; v[2] = v_in[2];
      continue;
    }
    if constexpr (EPI == kEpiNone) {
      if ((lane & 3) == 0) {
        double *dst = a.s.v_rounds + 3 * ((int64_t)r * a.n_pairs + pair);
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
      }
      n_es = r + 1;
      going = it_r >= kNewtonMaxIterations;
    }
  }
#ifdef PNEC_FRONT_DEBUG
  if (mine && (lane & 3) == 0) {
    atomicMax(&g_dbg[12], (unsigned long long)it);
    if (it >= kNewtonMaxIterations) { atomicAdd(&g_dbg[13], 1ull); atomicExch(&g_dbg[11], (unsigned long long)pair); }
    if (it >= 20) atomicAdd(&g_dbg[14], 1ull);
    atomicAdd(&g_dbg[15], (unsigned long long)it);
  }
#endif
  if constexpr (EPI == kEpiNone) {
    double M[9], w[3], V[9];
    es_value_grad<1>(Gs[quad], v, nullptr, M);
    sym_eig3(M, w, V);
    if (mine && (lane & 3) == 0) {
      a.s.n_es[pair] = n_es;
      a.s.its[pair] = it;
    }
    weighted_order_place(a.s, a.n_pairs, mine && (lane & 3) == 0, pair, w);
  } else {
    es_batch_translation_epilogue(a, Gs[quad], v, pair, mine, it, lane);
  }
}

constexpr int kHypPerRound = 16;  // hypotheses evaluated per round = quads per wavefront
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#include "pnec_es_schemes.inl"

// es_batch_kernel under eigensolver schemes 1 and 2 (pnec_es_schemes.inl): sixteen pairs per wavefront through the queue
// form (one task per quad).  kEpiTranslation: one minimisation and the same epilogue.  kEpiNone (the weighted stage):
// ALL of the stage's minimisations, chained -- the weights never change (C3), so round r + 1 minimises the same function
// from round r's result and none of it depends on the translation part of the rounds: `rounds` of them under scheme 1
// (each starts where the last one stopped short, through the rotation MATRIX as pnec.cc:310-311 hands it over), and under
// scheme 2 until one ends for another reason than maxfev (the rotation is final from there).
struct EsAltLds {
  double Gs[16][36];
  double tv[16][3], te[16][3];
  int tits[16], tflag[16], tlist[16];
};
template <int EPI, int SCHEME>
__global__ __launch_bounds__(kWave, 2) void es_batch_alt_kernel(const EsBatchArgs a, int rounds) {
  const int lane = threadIdx.x;
  const int quad = lane >> 2;
  __shared__ EsAltLds lds;
  const int64_t first_pair = 16 * (int64_t)blockIdx.x;
  for (int i = lane; i < 16 * 36; i += kWave) {
    int64_t p = first_pair + i / 36;
    p = p < a.n_pairs ? p : a.n_pairs - 1;
    lds.Gs[i / 36][i % 36] = a.s.G[36 * p + i % 36];
  }
  const int64_t left = a.n_pairs - first_pair;
  const int n_mine = left < 16 ? (int)left : 16;  // pairs of this wavefront
  const bool mine = quad < n_mine;
  const int64_t pair = mine ? first_pair + quad : a.n_pairs - 1;
  if ((lane & 3) == 0) {
    lds.tv[quad][0] = a.s.v0[3 * pair]; lds.tv[quad][1] = a.s.v0[3 * pair + 1]; lds.tv[quad][2] = a.s.v0[3 * pair + 2];
    lds.tlist[quad] = quad;
  }
  wave_lds_sync();
  int it_first = 0;
  if constexpr (EPI == kEpiTranslation) {
    es_minimise_queue_alt<SCHEME, kWkEsTailEvals>(n_mine, lds.tlist, lds.Gs, lds.tv, lds.te, lds.tits, lds.tflag);
    wave_lds_sync();
    it_first = lds.tits[quad];
  } else {
    bool going = mine;
    int n_es = 0;
    for (int r = 0; r < rounds; ++r) {  // wave-uniform exit below
      // the round's queue: the pairs whose rotation may still move
      const unsigned long long gb = __builtin_amdgcn_ballot_w64(going && (lane & 3) == 0);
      const int n_tasks = __builtin_popcountll(gb);
      if (n_tasks == 0) break;
      if (going && (lane & 3) == 0) {
        lds.tlist[__builtin_popcountll(gb & ((1ull << lane) - 1ull))] = quad;
        if (SCHEME == 1 && r > 0) {  // a new adapter holding rel_pose's rotation MATRIX (pnec.cc:310-311)
          double vv[3] = {lds.tv[quad][0], lds.tv[quad][1], lds.tv[quad][2]}, Rr[9];
          cayley_to_rot(vv, Rr);
          rot_to_cayley(Rr, vv);
          lds.tv[quad][0] = vv[0]; lds.tv[quad][1] = vv[1]; lds.tv[quad][2] = vv[2];
        }
      }
      wave_lds_sync();
      es_minimise_queue_alt<SCHEME, kWkEsFirstEvals>(n_tasks, lds.tlist, lds.Gs, lds.tv, lds.te, lds.tits, lds.tflag);
      wave_lds_sync();
      if (going) {
        if ((lane & 3) == 0) {
          double *dst = a.s.v_rounds + 3 * ((int64_t)r * a.n_pairs + pair);
          dst[0] = lds.tv[quad][0]; dst[1] = lds.tv[quad][1]; dst[2] = lds.tv[quad][2];
        }
        if (r == 0) it_first = lds.tits[quad];
        n_es = r + 1;
        if (SCHEME == 2 && lds.tflag[quad] == 0) going = false;
      }
      wave_lds_sync();
    }
    double M[9], w[3], V[9];
    {
      const double vq[3] = {lds.tv[quad][0], lds.tv[quad][1], lds.tv[quad][2]};
      es_value_grad<1>(lds.Gs[quad], vq, nullptr, M);
      sym_eig3(M, w, V);
    }
    if (mine && (lane & 3) == 0) {
      a.s.n_es[pair] = n_es;
      a.s.its[pair] = it_first;
    }
    weighted_order_place(a.s, a.n_pairs, mine && (lane & 3) == 0, pair, w);
    return;
  }
  double v[3] = {lds.tv[quad][0], lds.tv[quad][1], lds.tv[quad][2]};
  es_batch_translation_epilogue(a, lds.Gs[quad], v, pair, mine, it_first, lane);
}

// per-correspondence n = f1 x R f2 and B = f1hat R Sigma R' f1hat' + reg I (packed symmetric)
// (n, B) of one correspondence from its twelve payload values e = f1 | f2 | Sigma (xx xy xz yy yz zz)
__device__ __forceinline__ void corr_nb_of(const double (&e)[12], const double (&R)[9], double reg, double (&nn)[3],
                                           double (&B)[6]);
__device__ __forceinline__ void corr_nb(const double *base, int stride, int idx, const double (&R)[9],
                                        double reg, double (&nn)[3], double (&B)[6]) {
  double e[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) e[c] = base[(int64_t)c * stride + idx];
  corr_nb_of(e, R, reg, nn, B);
}
__device__ __forceinline__ void corr_nb_of(const double (&e)[12], const double (&R)[9], double reg, double (&nn)[3],
                                           double (&B)[6]) {
  const double f1[3] = {e[0], e[1], e[2]};
  const double f2[3] = {e[3], e[4], e[5]};
  const double S[6] = {e[6], e[7], e[8], e[9], e[10], e[11]};
  const double u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                       R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
  nn[0] = f1[1] * u[2] - f1[2] * u[1];
  nn[1] = f1[2] * u[0] - f1[0] * u[2];
  nn[2] = f1[0] * u[1] - f1[1] * u[0];
  // P = f1hat R: column c of P = f1 x (column c of R)
  double P[9];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double rx = R[c], ry = R[3 + c], rz = R[6 + c];
    P[c] = f1[1] * rz - f1[2] * ry;
    P[3 + c] = f1[2] * rx - f1[0] * rz;
    P[6 + c] = f1[0] * ry - f1[1] * rx;
  }
  double PS[9];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    PS[3 * r + 0] = P[3 * r] * S[0] + P[3 * r + 1] * S[1] + P[3 * r + 2] * S[2];
    PS[3 * r + 1] = P[3 * r] * S[1] + P[3 * r + 1] * S[3] + P[3 * r + 2] * S[4];
    PS[3 * r + 2] = P[3 * r] * S[2] + P[3 * r + 1] * S[4] + P[3 * r + 2] * S[5];
  }
  B[0] = PS[0] * P[0] + PS[1] * P[1] + PS[2] * P[2] + reg;
  B[1] = PS[0] * P[3] + PS[1] * P[4] + PS[2] * P[5];
  B[2] = PS[0] * P[6] + PS[1] * P[7] + PS[2] * P[8];
  B[3] = PS[3] * P[3] + PS[4] * P[4] + PS[5] * P[5] + reg;
  B[4] = PS[3] * P[6] + PS[4] * P[7] + PS[5] * P[8];
  B[5] = PS[6] * P[6] + PS[7] * P[7] + PS[8] * P[8] + reg;
}

// obj_fun for one direction over the whole pair (scf.cc:43-51), all lanes get the sum
__device__ double obj_fun_pair(const double *base, int n, int stride, const double (&R)[9], double reg,
                               const double (&t)[3], int lane) {
  double acc = 0.0;
  for (int idx = lane; idx < n; idx += kWave) {
    double nn[3], B[6];
    corr_nb(base, stride, idx, R, reg, nn, B);
    const double a = t[0] * nn[0] + t[1] * nn[1] + t[2] * nn[2];
    const double d = t[0] * (B[0] * t[0] + B[1] * t[1] + B[2] * t[2]) + t[1] * (B[1] * t[0] + B[3] * t[1] + B[4] * t[2]) +
                     t[2] * (B[2] * t[0] + B[4] * t[1] + B[5] * t[2]);
    acc = __builtin_fma(a * a, fast_rcp(d), acc);
  }
  return wave_allreduce_sum(acc);
}

// ---- PNEC::WeightedEigensolver ---------------------------------------------------------------
// RES: the pair has <= 512 correspondences, so each lane keeps n and B of its (<= 8)
// correspondences in registers for the 24 search batches and the SCF steps of a round instead of
// re-reading the payload and rebuilding them every time (they only change with the rotation).
#ifndef PNEC_WES_WAVES_PER_SIMD
#define PNEC_WES_WAVES_PER_SIMD 2
#endif
// WPP > 1 (resident form only): a pair of up to 512 WPP correspondences on WPP wavefronts, each keeping its
// share's tables in registers; every sum over the pair is a wavefront sum + one exchange through LDS (pair_sum
// below), everything else runs identically in all of them (same bits: all add the partial sums in the same
// order), so their control flow never parts and the barriers inside pair_sum are always met by all.
// The LDS of a block (one struct for whatever forms its kernel can run, so it is the largest, not the sum):
template <int WMAX>
struct WeightedLds {
  double cand[21][3];    // streaming form: the directions of a batch
  // exchange between the wavefronts of a pair: [buffer][wavefront][value]; the buffers alternate so that a
  // wavefront that runs ahead into the next exchange cannot overwrite what another has yet to read
  double xch[2][WMAX][8];
  // resident form: each wavefront's lower bounds of the 500 directions' costs for the current tables
  // (direction 64 j + lane at [j][lane])
  double lb[WMAX][8][kWave];
  unsigned long long cmask[8];   // directions that can still be chosen (bit l of word j: direction 64 j + l)
  float cost32[512];             // the rare pair with many of them: their single-precision costs
  float cost32p[WMAX > 1 ? WMAX : 1][WMAX > 1 ? 512 : 1];  // ... each wavefront's share of those
};
template <bool RES, int WPP, typename Lds>
__device__ __forceinline__ void weighted_pair(const FrontArgs &a, Lds &lds, const int64_t pair) {
  static_assert(WPP == 1 || RES, "several wavefronts per pair exist for the resident form only");
  const int lane = threadIdx.x & (kWave - 1);
  [[maybe_unused]] const int wave = threadIdx.x >> 6;
  const int n = a.count[pair];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const double *base = a.data + a.block_offset[pair];
  [[maybe_unused]] double (*cand)[3] = lds.cand;
  [[maybe_unused]] auto &xch = lds.xch;
  [[maybe_unused]] int xpar = 0;
  // the block's barrier (both wavefronts) or, for one wavefront, just the LDS fence
  auto pair_sync = [&]() {
    if constexpr (WPP > 1) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  };
  // sums over the whole pair of K values per lane (every lane of both wavefronts ends with the same bits)
  auto pair_sum = [&](auto &x /* double[K] */) {
    constexpr int K = sizeof(x) / sizeof(double);
#pragma unroll
    for (int k = 0; k < K; ++k) x[k] = wave_allreduce_sum(x[k]);
    if constexpr (WPP > 1) {
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) xch[xpar][wave][k] = x[k];
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double t = xch[xpar][0][k];
#pragma unroll
        for (int w = 1; w < WPP; ++w) t += xch[xpar][w][k];
        x[k] = t;
      }
      xpar ^= 1;
    }
  };
  // Everything the kernel reads per pair -- start pose, the first round's minimiser, its iteration count and the number
  // of rounds that have one (es_batch_kernel's results) -- is requested HERE, in one trip to memory: read where it was
  // first used, each was a dependent round trip of its own (~8 k clocks of the pair's 80 k: phase clocks, round 5).
  const double *iqp = a.init_q + 4 * pair, *itp = a.init_t + 3 * pair;
  const double *pvp = a.pre_v_rounds + 3 * pair;
  double q0[4] = {iqp[0], iqp[1], iqp[2], iqp[3]};
  const double t0[3] = {itp[0], itp[1], itp[2]};
  const double pre_v0[3] = {pvp[0], pvp[1], pvp[2]};
  const int pre_its0 = a.pre_its[pair];
  const int pre_n_es0 = a.pre_n_es[pair];
  {
    const double qn = 1.0 / sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    for (int k = 0; k < 4; ++k) q0[k] *= qn;
  }
  double R0[9];
  rot_from_quat(q0, R0);
  double R[9], t[3] = {t0[0], t0[1], t0[2]}, v[3] = {pre_v0[0], pre_v0[1], pre_v0[2]};
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = R0[i];
  unsigned long long ph_clk[kPhCount] = {0};
  const unsigned long long ph_start = a.trace ? __builtin_amdgcn_s_memtime() : 0ull;
  [[maybe_unused]] const unsigned long long ph_start_real = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  PNEC_PHASE_BEGIN();
  // (weights come from the INITIAL pose in every iteration (C3): the 36 weighted sums never change, so every eigenvalue
  // minimisation of the stage ran before this kernel, chained in es_batch_kernel; until round 5 the ones after the first
  // ran here -- a Newton iteration that one pair in thousands entered and every pair paid 224 B/lane of scratch for)
  PNEC_PHASE_END(kPhSums);

  constexpr int KR = RES ? 8 : 1;
  double rn[KR][3], rB[KR][6];
  // visit every correspondence of the lane with its (n, B): from registers or by streaming
  auto for_each_corr = [&](auto &&body) {
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < KR; ++k) body(rn[k], rB[k]);
    } else {
      for (int idx = lane; idx < n; idx += kWave) {
        double nn[3], B[6];
        corr_nb(base, stride, idx, R, a.reg, nn, B);
        body(nn, B);
      }
    }
  };

  double fib_min_cost = 0.0;
  int fib_min_idx = -1;  // -1: no stored search yet (streaming form) / no direction can beat the translation
  [[maybe_unused]] bool mlo_valid = false;  // resident form: lds.lb belongs to the current tables
  // resident form: what the stored search (fib_min_*) is good for while the tables stand.  search_global: it is the
  // smallest cost of ALL 500 directions (nothing left to find out); otherwise no direction costs less than
  // search_thr, the current translation's cost it was run against (nothing to find out while the cost stays below)
  [[maybe_unused]] bool search_global = false;
  [[maybe_unused]] double search_thr = -1.0;
  int first_iterations = 0;
  bool rotation_final = false;
  // One round of pnec.cc:295-346.  WITH_ES: the round starts with an eigensolver call (the rotation may
  // still move) and builds the (n, B) tables after it; without, the rotation is final, the tables of the
  // previous round are still in registers and only the translation part runs.  Returns true when the
  // round left t bit for bit where it found it.
  auto round = [&](int it, auto with_es) -> bool {
    constexpr bool WITH_ES = decltype(with_es)::value;
    // The weights never change (C3), so every round minimises the same function from the previous
    // optimum: once a call has ended for any reason other than the iteration cap, the rotation is
    // final and later rounds only redo the translation (newton = 0: "did not move").
    int newton = 0;
    if constexpr (WITH_ES) {
      if (it > 0 && it < pre_n_es0) {
        const double *pv = a.pre_v_rounds + 3 * ((int64_t)it * a.n_pairs + pair);
        v[0] = pv[0]; v[1] = pv[1]; v[2] = pv[2];
      }
      // (beyond the pair's last call -- the streaming form's later rounds -- the rotation stays: newton = 0)
      newton = it == 0 ? pre_its0 : (it < pre_n_es0 ? 1 : 0);
      rotation_final = it + 1 >= pre_n_es0;
    }
    if (it == 0) first_iterations = newton;
    PNEC_PHASE_END(kPhNewton);
    [[maybe_unused]] const bool same_rotation = (it > 0 && newton == 0 && fib_min_idx >= 0);
    if constexpr (WITH_ES) cayley_to_rot(v, R);
    if constexpr (RES && WITH_ES) {
      // built after the eigensolver call so that the arrays are dead across it (the Newton iteration
      // and 144 resident registers do not fit a wavefront's register file together).  No branches: the
      // loads of all eight correspondences are in flight together (clamped addresses for the lanes
      // beyond the pair), the padding entries are patched afterwards.
      mlo_valid = false;
      search_global = false;
      search_thr = -1.0;
      // (loads in the scalar-base form, pnec_device.hpp load_sets8x12_saddr: the compiler's own version of this
      // loop kept 96 precomputed 64-bit addresses in scratch and reloaded each in front of its load, one memory
      // round trip after the other -- this phase took 170 k clocks of the stage's 375 k)
      const unsigned long long b64 = reinterpret_cast<unsigned long long>(base) +
                                     (unsigned long long)(WPP > 1 ? wave : 0) * KR * kWave * sizeof(double);
      const char *sbase = reinterpret_cast<const char *>(
          ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32)) << 32) |
          (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64));
      const size_t plane_bytes = (size_t)(unsigned)__builtin_amdgcn_readfirstlane(stride) * sizeof(double);
      double pe[8][12];   // (this branch is the resident form: KR == 8)
      {
        // the sets of 128 correspondences of this wavefront's share that start inside the pair's planes (stride = n rounded
        // up to 64); slot s of lane l holds correspondence first + 128 (s / 2) + 2 l + (s & 1) (load_sets4x12x2_saddr)
        const int first = (WPP > 1 ? wave * KR * kWave : 0);
        const int sets = (stride - first + 2 * kWave - 1) / (2 * kWave);
        const unsigned nt = (unsigned)__builtin_amdgcn_readfirstlane(sets < 0 ? 0 : (sets > 4 ? 4 : sets));
        load_sets4x12x2_saddr(pe, sbase, plane_bytes, 16u * (unsigned)lane, nt);
      }
      if (threadIdx.x == 0) PNEC_WORK_ADD(kWkWesTableCorr, n);
      // the rotation as SCALARS here (it is one per pair; every product below takes one entry of it): this is where the
      // kernel has the fewest registers -- 96 payload values in flight, the tables forming -- and as eighteen vector
      // registers the rotation was spilled and reloaded from scratch for each of the eight correspondences (32 reloads
      // per build, each waited for)
      double Rs[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rs[i] = to_sgpr(R[i]);
#pragma unroll
      for (int k = 0; k < KR; ++k) {
        const int idx = (WPP > 1 ? wave * KR * kWave : 0) + 2 * kWave * (k / 2) + 2 * lane + (k & 1);
        const bool in = idx < n;
        corr_nb_of(pe[k], Rs, a.reg, rn[k], rB[k]);
        if (!in) {  // padding: contributes exactly 0 to every sum (n = 0, B = I)
          rn[k][0] = rn[k][1] = rn[k][2] = 0.0;
          rB[k][0] = rB[k][3] = rB[k][5] = 1.0;
          rB[k][1] = rB[k][2] = rB[k][4] = 0.0;
        }
      }
    }
    PNEC_PHASE_END(kPhTables);
    const double t_in[3] = {t[0], t[1], t[2]};
    // the current translation's cost: what a Fibonacci direction has to beat (pnec.cc:318-325)
    double cur_cost = 0.0;
    for_each_corr([&](const double(&nn)[3], const double(&B)[6]) {
      const double aa = t[0] * nn[0] + t[1] * nn[1] + t[2] * nn[2];
      const double d = t[0] * (B[0] * t[0] + B[1] * t[1] + B[2] * t[2]) + t[1] * (B[1] * t[0] + B[3] * t[1] + B[4] * t[2]) +
                       t[2] * (B[2] * t[0] + B[4] * t[1] + B[5] * t[2]);
      cur_cost = __builtin_fma(aa * aa, fast_rcp(d), cur_cost);
    });
    {
      double cs[1] = {cur_cost};
      pair_sum(cs);
      cur_cost = cs[0];
    }
    if (threadIdx.x == 0) PNEC_WORK_ADD(kWkWesCostCorr, n);
    PNEC_PHASE_END(kPhCost);
    [[maybe_unused]] auto energy_term = [](double tx, double ty, double tz, const double(&nn)[3], const double(&B)[6]) {
      const double aa = tx * nn[0] + ty * nn[1] + tz * nn[2];
      const double d = tx * (B[0] * tx + B[1] * ty + B[2] * tz) + ty * (B[1] * tx + B[3] * ty + B[4] * tz) +
                       tz * (B[2] * tx + B[4] * ty + B[5] * tz);
      return aa * aa * fast_rcp(d);
    };
    if constexpr (RES) {
      // The 500 Fibonacci directions (scf.cc:43-66, pnec.cc:313-325) through an EXACT bound instead of 500
      // evaluations.  The search only matters through "the first direction whose cost is the smallest AND strictly
      // below the current translation's" -- and every term of a direction's cost is
      //     (t.n_i)^2 / (t' B_i t)  >=  (t.n_i)^2 / trace(B_i)          (B_i is positive semi-definite, |t| = 1),
      // so  cost(t) >= t' M t  with  M = sum_i n_i n_i' / trace(B_i): one 3x3 matrix per rotation, six
      // multiply-adds per direction.  A direction whose bound is not below the current translation's cost
      // cannot be chosen, whatever its cost; the others (on the simulated and the KITTI-like data: none in
      // almost every pair -- the translation handed over by the eigensolver stage is already far better than
      // any grid point nine degrees from its neighbours) are evaluated in double precision, in index order, as
      // before.  Same decisions, same bits: until round 3 this was a packed single-precision pre-screen of all
      // 500 directions over the resident tables -- 28 k of the stage's 44 k instructions per pair, and 72
      // registers of single-precision tables.
      if (!mlo_valid) {
        double m[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < KR; ++k) {
          const double w = fast_rcp(rB[k][0] + rB[k][3] + rB[k][5]);
          const double wx = w * rn[k][0], wy = w * rn[k][1], wz = w * rn[k][2];
          m[0] = __builtin_fma(wx, rn[k][0], m[0]); m[1] = __builtin_fma(wx, rn[k][1], m[1]);
          m[2] = __builtin_fma(wx, rn[k][2], m[2]); m[3] = __builtin_fma(wy, rn[k][1], m[3]);
          m[4] = __builtin_fma(wy, rn[k][2], m[4]); m[5] = __builtin_fma(wz, rn[k][2], m[5]);
        }
        pair_sum(m);
        if (threadIdx.x == 0) PNEC_WORK_ADD(kWkWesBoundCorr, n);
        // t' M t for this lane's directions from the planes of products (all 48 loads in flight together), a hair
        // below the bound: the costs it is compared with carry their own rounding.  The bounds belong to the
        // tables: later rounds on the same rotation compare them with their own current cost without a load.
        // (the loads in the scalar-base form, like the tables'; half of the grid at a time: 48 doubles in flight on
        // top of the 144 table registers do not fit the file -- left to the compiler, the 48 addresses were computed
        // at kernel entry, parked in scratch and fetched back one by one)
        {
          const char *pb = reinterpret_cast<const char *>(&c_fib_prod[0][0]);
          const unsigned voff = 8u * (unsigned)lane;
          auto half = [&](auto hc) {
            constexpr int j0 = 4 * decltype(hc)::value;
            double fp[4][6];
            fib_prod_load4<j0>(fp, pb, voff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
              lds.lb[WPP > 1 ? wave : 0][j0 + j][lane] = (m[0] * fp[j][0] + m[3] * fp[j][1] + m[5] * fp[j][2] + m[1] * fp[j][3] +
                                                          m[2] * fp[j][4] + m[4] * fp[j][5]) * (1.0 - 1.0e-9);
          };
          half(std::integral_constant<int, 0>{});
          half(std::integral_constant<int, 1>{});
        }
        mlo_valid = true;
      }
      // The stored search stands while the tables do: either it found the smallest of all 500 costs, or it showed
      // that none is below search_thr and the current translation still costs no more than that.  (A NaN cost:
      // no comparison with it is true, the search result cannot be used -- nothing to do either.)
      const bool need_search = !search_global && !(cur_cost <= search_thr) && cur_cost == cur_cost;
      if (need_search) {
        fib_min_idx = -1;
        // which directions can still be chosen (written so that a NaN bound keeps its direction: nothing is pruned
        // on garbage); the masks are the same in every wavefront of the pair
        int n_cand = 0;
        bool full_grid = false;  // the search below went over all 500 directions
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned long long cm =
              __builtin_amdgcn_ballot_w64(kWave * j + lane < 500 && !(lds.lb[WPP > 1 ? wave : 0][j][lane] >= cur_cost));
          n_cand += __builtin_popcountll(cm);
          if (lane == 0 && (WPP == 1 || wave == 0)) lds.cmask[j] = cm;
        }
        if (a.trace) {  // diagnostics: searches run, candidates they started with
          ph_clk[8] += 1;
          ph_clk[9] += (unsigned long long)n_cand;
        }
        pair_sync();
        // one direction's cost in double precision, with the arithmetic of the current translation's cost (the
        // comparison between the two decides whether the search result is used at all); candidates come in index
        // order: ties keep the first
        auto exact = [&](int c) {
          const double *fc = c_fib + kFibStride * c;
          const double tx = fc[0], ty = fc[1], tz = fc[2];
          double sacc = 0.0;
#pragma unroll
          for (int k = 0; k < KR; ++k) {
            const double aa = tx * rn[k][0] + ty * rn[k][1] + tz * rn[k][2];
            const double d = tx * (rB[k][0] * tx + rB[k][1] * ty + rB[k][2] * tz) +
                             ty * (rB[k][1] * tx + rB[k][3] * ty + rB[k][4] * tz) +
                             tz * (rB[k][2] * tx + rB[k][4] * ty + rB[k][5] * tz);
            sacc = __builtin_fma(aa * aa, fast_rcp(d), sacc);
          }
          double cs[1] = {sacc};
          pair_sum(cs);
          if (threadIdx.x == 0) PNEC_WORK_ADD(kWkWesCostCorr, n);
          if (fib_min_idx < 0 || cs[0] < fib_min_cost) {
            fib_min_cost = cs[0];
            fib_min_idx = c;
          }
          if (a.trace) ph_clk[10] += 1;  // diagnostics: double-precision evaluations
        };
#ifdef PNEC_WES_NO_PRESCREEN   // A/B: every candidate straight to double precision (what the common path costs alone)
        if (n_cand > 0) {
#else
        if (n_cand > 0 && n_cand <= 12) {
#endif
          // the usual case when there are any: a handful -- straight to double precision, in index order
          for (int j = 0; j < 8; ++j) {
            unsigned long long cand = lds.cmask[j];
            while (cand != 0ull) {
              exact(kWave * j + (int)__builtin_ctzll(cand));
              cand &= cand - 1ull;
            }
          }
        }
#ifndef PNEC_WES_NO_PRESCREEN
        else if (n_cand > 0) {
          full_grid = true;
          // The rare pair whose translation is barely observable (its cost is nearly flat over the sphere, so the
          // bound prunes little; 1-2 % of the simulated pairs, none of the KITTI-like ones): ALL 500 directions go
          // through the packed single-precision pre-screen this stage used for every pair until round 3
          // (v_pk_fma_f32: two correspondences per lane and instruction, no Newton steps on the reciprocals, ~1e-6
          // relative; direction c + r ends up in row r), and every direction within 1e-4 of the smallest single-
          // precision cost is evaluated again in double precision, in index order.  All of them, not just the
          // candidates: what comes out is the smallest cost of the whole grid, which stands whatever the translation
          // does in later rounds (such a pair's cost wanders up and down, and a search per round took the longest
          // pair of a 20 000-pair launch to 4 M clocks -- the launch's whole duration).
          const int my_row = lane >> 4;
          for (int c = 0; c < 500; c += 4) {
            // (the single-precision copies of the tables are made here, two correspondences at a time, and die here:
            // kept resident they cost the common path 72 registers it does not have)
            v2f acc4[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
            // (an opaque zero added before the conversions: they do not depend on the direction, so the compiler moved
            // all 72 of them out of this loop AND out of the rounds' loop, i.e. in front of this rare branch, into the
            // path of every pair -- 36 packed values converted and spilled to scratch per pair, for a loop that 1-2 %
            // of the pairs enter)
            double zero_here = 0.0;
            asm volatile("" : "+v"(zero_here));
#pragma unroll
            for (int j = 0; j < KR / 2; ++j) {
              v2f fn[3], fB[6];
#pragma unroll
              for (int cc = 0; cc < 3; ++cc) fn[cc] = v2f{(float)(rn[2 * j][cc] + zero_here), (float)(rn[2 * j + 1][cc] + zero_here)};
#pragma unroll
              for (int cc = 0; cc < 6; ++cc) fB[cc] = v2f{(float)(rB[2 * j][cc] + zero_here), (float)(rB[2 * j + 1][cc] + zero_here)};
#pragma unroll
              for (int jd = 0; jd < 4; ++jd) {
                const float *fc = c_fib32 + kFibStride * (c + jd);
                v2f d = fB[0] * fc[3];
                d = fB[3] * fc[4] + d;
                d = fB[5] * fc[5] + d;
                d = fB[1] * fc[6] + d;
                d = fB[2] * fc[7] + d;
                d = fB[4] * fc[8] + d;
                v2f aa = fn[0] * fc[0];
                aa = fn[1] * fc[1] + aa;
                aa = fn[2] * fc[2] + aa;
                const v2f y = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                acc4[jd] = (aa * aa) * y + acc4[jd];
              }
            }
            float s4[4];
#pragma unroll
            for (int jd = 0; jd < 4; ++jd) s4[jd] = acc4[jd].x + acc4[jd].y;
            const float b0 = swap_add32_f(s4[0], s4[2]), b1 = swap_add32_f(s4[1], s4[3]);
            const float sum = row_allreduce_sum_f(swap_add16_f(b0, b1));  // row r: direction c + r
            if ((lane & 15) == 0) {
              if constexpr (WPP > 1) lds.cost32p[wave][c + my_row] = sum;
              else lds.cost32[c + my_row] = sum;
            }
          }
          if constexpr (WPP > 1) {  // the wavefronts' shares of every direction's cost, added once
            __syncthreads();
            for (int i = (int)threadIdx.x; i < 500; i += WPP * kWave) {
              float t32 = lds.cost32p[0][i];
#pragma unroll
              for (int w = 1; w < WPP; ++w) t32 += lds.cost32p[w][i];
              lds.cost32[i] = t32;
            }
          }
          pair_sync();
          float m32 = __builtin_inff();
          for (int i = lane; i < 500; i += kWave) m32 = fminf(m32, lds.cost32[i]);  // fminf skips NaN
          m32 = wave_allreduce_min_f(m32);
          const float thr32 = m32 + 1.0e-4f * fabsf(m32);
          for (int i0 = 0; i0 < 500; i0 += kWave) {
            const int i = i0 + lane;
            unsigned long long cand = __builtin_amdgcn_ballot_w64(i < 500 && lds.cost32[i < 500 ? i : 0] <= thr32);
            while (cand != 0ull) {
              exact(i0 + (int)__builtin_ctzll(cand));
              cand &= cand - 1ull;
            }
          }
        }
#endif
        // what this search is good for from here on (see above)
        if (fib_min_idx >= 0 && (fib_min_cost < cur_cost || full_grid)) {
          search_global = true;
        } else {
          fib_min_idx = -1;
          search_thr = cur_cost;
        }
      }
    } else if (!same_rotation) {
      fib_min_idx = -1;
        // streaming: 21 directions at a time, per correspondence n, B rebuilt once per batch
        for (int b0 = 0; b0 < 500; b0 += 21) {
          const int nb = (500 - b0 < 21) ? 500 - b0 : 21;
          if (lane < nb * 3) cand[lane / 3][lane % 3] = c_fib[kFibStride * (b0 + lane / 3) + lane % 3];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          double acc[kNumAcc];
#pragma unroll
          for (int c = 0; c < kNumAcc; ++c) acc[c] = 0.0;
          for_each_corr([&](const double(&nn)[3], const double(&B)[6]) {
#pragma unroll
            for (int c = 0; c < kNumAcc; ++c) acc[c] += energy_term(cand[c][0], cand[c][1], cand[c][2], nn, B);
          });
          double sums[kNumAcc];
          wave_reduce21(acc, sums);
#pragma unroll
          for (int c = 0; c < kNumAcc; ++c) {
            if (c < nb && (fib_min_idx < 0 || sums[c] < fib_min_cost)) {
              fib_min_cost = sums[c];
              fib_min_idx = b0 + c;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        }
    }
    PNEC_PHASE_END(kPhSearch);
    if (fib_min_idx >= 0 && fib_min_cost < cur_cost) {
      t[0] = c_fib[kFibStride * fib_min_idx];
      t[1] = c_fib[kFibStride * fib_min_idx + 1];
      t[2] = c_fib[kFibStride * fib_min_idx + 2];
    }
    // scf: 10 steps of  t <- eigenvector of the smallest eigenvalue of sum A_i / (t' B_i t)
    for (int step = 0; step < 10; ++step) {
      double e[6] = {0, 0, 0, 0, 0, 0};
      for_each_corr([&](const double(&nn)[3], const double(&B)[6]) {
        const double d = t[0] * (B[0] * t[0] + B[1] * t[1] + B[2] * t[2]) + t[1] * (B[1] * t[0] + B[3] * t[1] + B[4] * t[2]) +
                         t[2] * (B[2] * t[0] + B[4] * t[1] + B[5] * t[2]);
        const double w = fast_rcp(d);
        e[0] += w * nn[0] * nn[0]; e[1] += w * nn[0] * nn[1]; e[2] += w * nn[0] * nn[2];
        e[3] += w * nn[1] * nn[1]; e[4] += w * nn[1] * nn[2]; e[5] += w * nn[2] * nn[2];
      });
      pair_sum(e);
      if (threadIdx.x == 0) PNEC_WORK_ADD(kWkWesScfCorr, n);
      const double E[9] = {e[0], e[1], e[2], e[1], e[3], e[4], e[2], e[4], e[5]};
      // smallest eigenvector of E: Rayleigh-quotient iteration from the current t (the iteration is
      // near its fixed point after the first step or two), Jacobi sweeps when that cannot be vouched for
      double tn[3] = {t[0], t[1], t[2]}, lam;
      if (!sym_eig3_min_rqi(E, tn, lam)) {
        double w3[3], V[9];
        sym_eig3(E, w3, V);
        tn[0] = V[0]; tn[1] = V[3]; tn[2] = V[6];
      }
      {  // the decomposition's sign convention: largest-magnitude component positive
        double big = tn[0];
        if (fabs(tn[1]) > fabs(big)) big = tn[1];
        if (fabs(tn[2]) > fabs(big)) big = tn[2];
        if (big < 0.0) { tn[0] = -tn[0]; tn[1] = -tn[1]; tn[2] = -tn[2]; }
      }
      const double moved = fmax(fabs(tn[0] - t[0]), fmax(fabs(tn[1] - t[1]), fabs(tn[2] - t[2])));
      t[0] = tn[0]; t[1] = tn[1]; t[2] = tn[2];
      // the iteration has reached its fixed point to rounding (a few ulp): the remaining steps of
      // the reference's fixed count of 10 would only reproduce that noise
      if (moved <= 4e-15) break;
    }
    PNEC_PHASE_END(kPhScf);
    return t[0] == t_in[0] && t[1] == t_in[1] && t[2] == t_in[2];
  };
  const int rounds = a.weighted_iterations - 1;
  int it = 0;
  {
    // (A) rounds that may still move the rotation (normally just the first: the call converges)
    while (it < rounds && !rotation_final) {
      round(it, std::true_type{});
      ++it;
    }
    // (B) the rotation is final: the remaining rounds only redo the translation from the same tables.
    // A round that returns t unchanged bit for bit has the same inputs as the next one will have, so
    // every later round would reproduce it: stop there (exact, not a tolerance).
    for (; it < rounds; ++it) {
      bool unchanged;
      if constexpr (RES) unchanged = round(it, std::false_type{});
      else unchanged = round(it, std::true_type{}), rotation_final = true;
      if (unchanged) break;
    }
  }
  if (lane == 0 && (WPP == 1 || wave == 0)) {
    double qo[4];
    quat_from_rot_dev(R, qo);
    const double qn = 1.0 / sqrt(qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3]);
    for (int k = 0; k < 4; ++k) a.out_q[4 * pair + k] = qo[k] * qn;
    for (int k = 0; k < 3; ++k) a.out_t[3 * pair + k] = t[k];
    if (a.out_iterations) a.out_iterations[pair] = first_iterations;
    if (a.trace) {
      ph_clk[kPhTotal] = __builtin_amdgcn_s_memtime() - ph_start;
      ph_clk[7] = (ph_start_real & 0xffffffffull) | (__builtin_amdgcn_s_memrealtime() << 32);  // start | end, 100 MHz
      for (int k = 0; k < kPhCount; ++k) a.trace[kPhCount * pair + k] = ph_clk[k];
    }
  }
}

// A batch whose largest pair fits the resident form runs it for every pair, one wavefront per pair.  A ragged
// batch with larger pairs decides PER PAIR (the KITTI-like stream, 265..700 correspondences, ran the streaming form
// for all 23 190 pairs because 40 % of them exceed 512 BEFORE the inliers are extracted -- 8.2 of the chain's
// 11.6 ms): WMAX wavefronts per block; a pair of <= 512 correspondences runs the resident form on the first (the
// others leave at once), up to 1024 on two, 2048 on four, 4096 on eight, and only beyond that the streaming form
// (4 000 pairs x 2 048: 14.5 ms streaming).
template <bool RES>
__global__ __launch_bounds__(kWave, PNEC_WES_WAVES_PER_SIMD) void weighted_eigensolver_kernel(const FrontArgs a) {
  __shared__ WeightedLds<1> lds;
  weighted_pair<RES, 1>(a, lds, a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x);
}
template <int WMAX>
__global__ __launch_bounds__(WMAX *kWave, PNEC_WES_WAVES_PER_SIMD) void weighted_eigensolver_mixed_kernel(const FrontArgs a) {
  __shared__ WeightedLds<WMAX> lds;
  const int64_t pair = a.order ? (int64_t)a.order[blockIdx.x] : (int64_t)blockIdx.x;
  const int n = a.count[pair];
  const int wave = threadIdx.x >> 6;
  if (n <= 8 * kWave) {
    if (wave < 1) weighted_pair<true, 1>(a, lds, pair);
  } else if (n <= 16 * kWave) {
    if (wave < 2) weighted_pair<true, 2>(a, lds, pair);
  } else if (WMAX >= 4 && n <= 32 * kWave) {
    if constexpr (WMAX >= 4) {
      if (wave < 4) weighted_pair<true, 4>(a, lds, pair);
    }
  } else if (WMAX >= 8 && n <= 64 * kWave) {
    if constexpr (WMAX >= 8) weighted_pair<true, 8>(a, lds, pair);
  } else {
    if (wave < 1) weighted_pair<false, 1>(a, lds, pair);
  }
}

// ---- RANSAC around the eigensolver (pnec.cc:239-272; opengv::sac::Ransac<EigensolverSacProblem>
// restated from its published behaviour: hypotheses from `sample_size` random correspondences with
// the start rotation jittered by +-0.01 in Cayley space, score = (1 - f1.reproj1) + (1 - f2.reproj2)
// of the midpoint triangulation, inlier if score < threshold, adaptive bound
// k = log(1 - 0.99) / log(1 - w^s), eigensolver re-run on the inliers; rand() replaced by a
// counter-based hash of (seed, pair, hypothesis, draw)) ------------------------------------------
// (kernel shape: see ransac2_eigensolver_kernel below)
struct RansacArgs {
  const double *data;
  const int64_t *block_offset;
  const int64_t *offsets;  // AoS offsets (inlier mask is written in the caller's correspondence order)
  const int32_t *count;
  const double *init_q;
  double *out_q, *out_t;
  uint8_t *out_mask;
  int32_t *out_count, *out_iterations;
  unsigned long long *trace;  // null, or [n_pairs, 8] clocks per phase (PNEC_HIP_TRACE_FRONT)
  unsigned long long seed;
  unsigned long long pair_id_base;  // pair p draws as pair pair_id_base + p (shards of a larger set)
  int64_t n_pairs;
  int max_iterations, sample_size;
  double threshold;
  FrontScratch scratch;  // what es_batch_kernel needs to finish the pair (sums of the inliers, start, first inlier)
  // InlierExtraction fused into the pair's last pass (null: none): the target batch has the source's block layout
  int nc;                      // component planes of the source (6 NEC, 12 TARGET / HOST, 18 SYM)
  double *sel_data;
  const int64_t *sel_block;
  int32_t *sel_count;
  int64_t *sel_single_offsets; // a batch of ONE pair: its AoS offsets [0, m]
  // two-pair form: the pairs in launch order (null: as they lie in the batch); see ransac_order_kernel
  const int32_t *order;
  // two launches (PHASE 1 / 2 of ransac2_eigensolver_kernel): where the sequential rule of a pair stands after its first
  // round -- it, best count, k, the best model (R | t) -- and the list of the pairs that go on
  double *st_k, *st_it, *st_best, *st_model;   // [P], [P], [P], [P,12]
  int32_t *st_list, *st_count;                 // [kStBuckets, P], [kStBuckets]
  int st_buckets;                              // buckets in use (1: the list unordered -- A/B)
  // PNEC_HIP_RANSAC_CHAINED_STARTS: every hypothesis starts from the rotation of the last model SCORED (+ jitter), as
  // opengv's adapter side effect has it [EXT]: a sequential dependence, so a round is ONE hypothesis per pair
  int chained;
  // two-pair form: blocks [0, n_double) take the pairs 2 b, 2 b + 1 (of the launch order), blocks from n_double on ONE
  // pair each, 2 n_double + (b - n_double): the launch's last wavefronts are short ones (see ransac_tail_singles)
  int64_t n_double;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double rng_uniform(unsigned long long seed, unsigned long long pair,
                                              unsigned long long hyp, unsigned long long draw) {
  const unsigned long long h =
      splitmix64(splitmix64(splitmix64(seed ^ 0xD1B54A32D192ED03ull) + pair) + (hyp << 20) + draw);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double reprojection_score(const double (&f1)[3], const double (&f2)[3],
                                                     const double (&R)[9], const double (&t)[3]) {
  const double u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                       R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
  const double b0 = t[0] * f1[0] + t[1] * f1[1] + t[2] * f1[2], b1 = t[0] * u[0] + t[1] * u[1] + t[2] * u[2];
  const double a00 = f1[0] * f1[0] + f1[1] * f1[1] + f1[2] * f1[2];
  const double a10 = f1[0] * u[0] + f1[1] * u[1] + f1[2] * u[2];
  const double a01 = -a10, a11 = -(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  const double inv_det = fast_rcp(a00 * a11 - a01 * a10);
  const double l0 = (a11 * b0 - a01 * b1) * inv_det, l1 = (-a10 * b0 + a00 * b1) * inv_det;
  double p[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p[k] = 0.5 * (l0 * f1[k] + t[k] + l1 * u[k]);
    d[k] = p[k] - t[k];
  }
  // the point in frame 2 is p2 = R'(p - t) = R'd; the score needs |p2| and f2 . p2 only, and a rotation keeps both:
  // |R'd| = |d|, f2 . R'd = (R f2) . d = u . d -- nine products less per correspondence and model (the checker forms p2;
  // the two differ in the last bits of a score that is compared with a threshold ten orders of magnitude above them)
  const double in1 = fast_rsqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const double in2 = fast_rsqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  return (1.0 - (f1[0] * p[0] + f1[1] * p[1] + f1[2] * p[2]) * in1) +
         (1.0 - (u[0] * d[0] + u[1] * d[1] + u[2] * d[2]) * in2);
}

// ---- RANSAC scoring: the pair's bearings in registers, one model after the other -------------------------------------
// The first kScoreTiles x 64 correspondences of the pair (f1 | f2: six planes) are loaded ONCE per round into registers,
// lane l holding correspondence 64 i + l of tile i (coalesced; scalar plane base + lane offset + immediate, see
// load_planes_saddr); what a larger pair has beyond them is streamed from memory per model.
constexpr int kScoreTiles = 8;
struct ScoreTiles { double f[kScoreTiles][6]; };

// ONE asm statement issues all 48 loads AND waits for them (why: pnec_device.hpp load_planes_saddr).  Tiles beyond the pair are branched over, wave-uniformly (a tile that starts
// inside the pair lies inside its planes: stride = n rounded up to 64, the padding reads zeros; one that starts beyond
// it may lie beyond the allocation), their registers keep the zeros they came with.  Scalar-base form of global_load
// (base + 32-bit lane offset + immediate; why: pnec_device.hpp load_planes_saddr); the s_nop covers a plane base the
// compiler has just fetched with a VALU instruction (VALU-written SGPR -> VMEM address: 5 wait states).
#define PNEC_ST_TILE(i, off)                                                          \
  "s_cmp_le_u32 %[nt], " #i "\n\ts_cbranch_scc1 1f\n\t"                               \
  "global_load_dwordx2 %[d" #i "0], %[lo], %[b0] offset:" #off "\n\t"                  \
  "global_load_dwordx2 %[d" #i "1], %[lo], %[b1] offset:" #off "\n\t"                  \
  "global_load_dwordx2 %[d" #i "2], %[lo], %[b2] offset:" #off "\n\t"                  \
  "global_load_dwordx2 %[d" #i "3], %[lo], %[b3] offset:" #off "\n\t"                  \
  "global_load_dwordx2 %[d" #i "4], %[lo], %[b4] offset:" #off "\n\t"                  \
  "global_load_dwordx2 %[d" #i "5], %[lo], %[b5] offset:" #off "\n\t"
#define PNEC_ST_OUT(i)                                                                                              \
  [d##i##0] "+v"(P.f[i][0]), [d##i##1] "+v"(P.f[i][1]), [d##i##2] "+v"(P.f[i][2]), [d##i##3] "+v"(P.f[i][3]), \
  [d##i##4] "+v"(P.f[i][4]), [d##i##5] "+v"(P.f[i][5])
__device__ __forceinline__ void score_tiles_load(ScoreTiles &P, const double *bs, int st, int nn, int lane) {
  static_assert(kScoreTiles == 8 && kWave == 64, "eight tiles of 64 correspondences, 512 bytes per tile and plane");
  const unsigned long long b64 = reinterpret_cast<unsigned long long>(bs);
  const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
  const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
  const char *sb = reinterpret_cast<const char *>(((unsigned long long)bhi << 32) | blo);
  const size_t plane_bytes = (size_t)(unsigned)__builtin_amdgcn_readfirstlane(st) * sizeof(double);
  const unsigned voff = 8u * (unsigned)lane;
  const unsigned nt = (unsigned)__builtin_amdgcn_readfirstlane((nn + kWave - 1) / kWave);  // tiles that start inside the pair
#pragma unroll
  for (int i = 0; i < kScoreTiles; ++i)
#pragma unroll
    for (int c = 0; c < 6; ++c) P.f[i][c] = 0.0;
  asm volatile("s_nop 4\n\t"
               PNEC_ST_TILE(0, 0) PNEC_ST_TILE(1, 512) PNEC_ST_TILE(2, 1024) PNEC_ST_TILE(3, 1536)
               PNEC_ST_TILE(4, 2048) PNEC_ST_TILE(5, 2560) PNEC_ST_TILE(6, 3072) PNEC_ST_TILE(7, 3584)
               "1:\n\ts_waitcnt vmcnt(0)"
               : PNEC_ST_OUT(0), PNEC_ST_OUT(1), PNEC_ST_OUT(2), PNEC_ST_OUT(3), PNEC_ST_OUT(4), PNEC_ST_OUT(5), PNEC_ST_OUT(6),
                 PNEC_ST_OUT(7)
               : [lo] "v"(voff), [nt] "s"(nt), [b0] "s"(sb), [b1] "s"(sb + plane_bytes), [b2] "s"(sb + 2 * plane_bytes),
                 [b3] "s"(sb + 3 * plane_bytes), [b4] "s"(sb + 4 * plane_bytes), [b5] "s"(sb + 5 * plane_bytes)
               : "memory", "scc");
}
#undef PNEC_ST_TILE
#undef PNEC_ST_OUT

// Inliers of ONE model over the whole pair.  `beat` is the count the model has to exceed to matter (the best so far of
// RANSAC's sequential rule): once even "every remaining correspondence is an inlier" cannot get it there, the scan stops
// and the count so far is returned -- any value <= beat leads to the same decision, so the rule's outcome is exactly the
// full scan's.  All wave-uniform.
// (Until round 3 the sixteen hypotheses of a round were scored side by side, a quad each, from tiles staged in LDS:
// nothing could be skipped, because a lane that leaves a SIMD instruction early saves nothing.  One after the other, a
// model fitted to a contaminated sample -- a handful of inliers against ~450 -- is dropped after one or two of the
// eight tiles.)
__device__ __forceinline__ int model_inliers_until_beaten(const ScoreTiles &P, const double *__restrict__ bs, int st, int nn,
                                                           const double (&R)[9], const double (&t)[3], double threshold,
                                                           int lane, int beat) {
  int cnt = 0;
  bool beaten = false;
  [[maybe_unused]] int tiles_scored = 0;
  auto tile = [&](auto tc) {
    constexpr int i = decltype(tc)::value;
    if (beaten || kWave * i >= nn) return;  // wave-uniform
#ifdef PNEC_WORK_COUNT
    ++tiles_scored;
#endif
    const double f1[3] = {P.f[i][0], P.f[i][1], P.f[i][2]}, f2[3] = {P.f[i][3], P.f[i][4], P.f[i][5]};
    const bool in = (kWave * i + lane < nn) && reprojection_score(f1, f2, R, t) < threshold;
    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
    const int left = nn - kWave * (i + 1);
    beaten = cnt + (left > 0 ? left : 0) <= beat;
  };
  tile(std::integral_constant<int, 0>{}); tile(std::integral_constant<int, 1>{});
  tile(std::integral_constant<int, 2>{}); tile(std::integral_constant<int, 3>{});
  tile(std::integral_constant<int, 4>{}); tile(std::integral_constant<int, 5>{});
  tile(std::integral_constant<int, 6>{}); tile(std::integral_constant<int, 7>{});
#ifdef PNEC_WORK_COUNT
  if (lane == 0) PNEC_WORK_ADD(kWkRansacTiles, tiles_scored + ((beaten || nn <= kWave * kScoreTiles) ? 0 : (nn - kWave * kScoreTiles + kWave - 1) / kWave));  // (the streamed rest: counted as scored in full)
#endif
  if (beaten || nn <= kWave * kScoreTiles) return cnt;
  // the rest of a pair of more than 512 correspondences: from memory, the next tile in flight while this one is scored
  double cur[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) cur[c] = bs[(int64_t)c * st + kWave * kScoreTiles + lane];
  for (int i0 = kWave * kScoreTiles; i0 < nn; i0 += kWave) {
    double nxt[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (i0 + kWave < nn) {
#pragma unroll
      for (int c = 0; c < 6; ++c) nxt[c] = bs[(int64_t)c * st + i0 + kWave + lane];
    }
    const double f1[3] = {cur[0], cur[1], cur[2]}, f2[3] = {cur[3], cur[4], cur[5]};
    const bool in = (i0 + lane < nn) && reprojection_score(f1, f2, R, t) < threshold;
    cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
    const int left = nn - i0 - kWave;
    if (cnt + (left > 0 ? left : 0) <= beat) break;
#pragma unroll
    for (int c = 0; c < 6; ++c) cur[c] = nxt[c];
  }
  return cnt;
}

// w^s for the rule's bound k = log(1 - 0.99) / log(1 - w^s), s = the sample size (an integer <= 16): by squaring.  (The
// library's pow() is a double-double logarithm and exponential -- ~250 instructions per new best model -- and the compiler
// kept its polynomial's coefficients in vector registers outside the rounds' loop, spilled them, and reloaded them one
// dependent round trip to scratch at a time inside the scoring loop: sixteen of them per call.  The result differs from
// pow()'s in the last bit at most; k is compared with integers it is never that close to.)
__device__ __forceinline__ double pow_sample(double w, int s) {
  double r = 1.0, b = w;
  for (int e = s; e > 0; e >>= 1) {
    if (e & 1) r *= b;
    b *= b;
  }
  return r;
}
enum : int { kRpSample = 0, kRpNewton, kRpModel, kRpScore, kRpConsume, kRpInliers, kRpFinal, kRpTotal };

// sum over the four lanes of a quad, every lane ends with it (two DPP butterflies; the same bits in all four)
__device__ __forceinline__ double quad_sum(double x) {
  x += dpp_perm<0xB1>(x);  // quad_perm [1,0,3,2]
  x += dpp_perm<0x4E>(x);  // quad_perm [2,3,0,1]
  return x;
}
__device__ __forceinline__ int quad_sum_int(int x) {
  x += __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  return x;
}

// ---- a hypothesis' sample and its sums, in registers (round 4) ------------------------------------------------------------
// The draw-until-distinct loop of the kernels' first version kept the sample in LDS (a dynamically indexed array is not
// registers) and read it back entry by entry in the duplicate test -- a chain of dependent LDS round trips per draw, and a
// wavefront fence per accepted entry in the two-pair form.  Here the sample is sixteen registers with static indices only:
// the duplicate test compares against all of them under a mask, the insertion is a select per register.  Same draws, same
// order, same sample.  (All four lanes of a quad draw the same sample.)
__device__ __forceinline__ void ransac_sample_regs(unsigned long long seed, unsigned long long pid, unsigned long long hh,
                                                   int n, int ss, int (&s)[PNEC_HIP_MAX_RANSAC_SAMPLE]) {
#pragma unroll
  for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) s[j] = -1;
  int m = 0;
  unsigned long long draw = 0;
  while (m < ss) {
    long long idx = (long long)(rng_uniform(seed, pid, hh, draw++) * (double)n);
    if (idx >= n) idx = n - 1;
    bool dup = false;
#pragma unroll
    for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) dup = dup || s[j] == (int)idx;  // (entries from m on are -1, idx >= 0)
    if (!dup) {
#pragma unroll
      for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) s[j] = (j == m) ? (int)idx : s[j];
      ++m;
    }
  }
}
// entry 4 t + role of the sample (what lane `role` of the quad works on in its t-th turn)
__device__ __forceinline__ int ransac_sample_pick(const int (&s)[PNEC_HIP_MAX_RANSAC_SAMPLE], int t, int role) {
  static_assert(PNEC_HIP_MAX_RANSAC_SAMPLE == 16, "four turns of four lanes");
  const int a0 = t == 0 ? s[0] : (t == 1 ? s[4] : (t == 2 ? s[8] : s[12]));
  const int a1 = t == 0 ? s[1] : (t == 1 ? s[5] : (t == 2 ? s[9] : s[13]));
  const int a2 = t == 0 ? s[2] : (t == 1 ? s[6] : (t == 2 ? s[10] : s[14]));
  const int a3 = t == 0 ? s[3] : (t == 1 ? s[7] : (t == 2 ? s[11] : s[15]));
  return role == 0 ? a0 : (role == 1 ? a1 : (role == 2 ? a2 : a3));
}
// The 36 sums G_kl[a][c] and the sum of f1 over the sample: the quad's lanes split it (lane `role` takes entries role,
// role + 4, ...), ALL of a lane's gathers in flight together (the first version waited out a round trip to memory per entry),
// accumulated entry by entry in the first version's order, then added up over the quad.
__device__ __forceinline__ void ransac_sample_sums(const double *bs, int st, int ss, bool active,
                                                   const int (&s)[PNEC_HIP_MAX_RANSAC_SAMPLE], int role, double (&Gl)[36],
                                                   double (&ev1)[3]) {
  constexpr int kTurns = PNEC_HIP_MAX_RANSAC_SAMPLE / 4;
  double e[kTurns][6];
  bool on[kTurns];
#pragma unroll
  for (int t = 0; t < kTurns; ++t) {
    on[t] = active && 4 * t + role < ss;
    const int idx = on[t] ? ransac_sample_pick(s, t, role) : 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) e[t][c] = on[t] ? bs[(int64_t)c * st + idx] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) ev1[c] = 0.0;
#pragma unroll
  for (int i = 0; i < 36; ++i) Gl[i] = 0.0;
#pragma unroll
  for (int t = 0; t < kTurns; ++t) {
    if (on[t]) {
      const double f1[3] = {e[t][0], e[t][1], e[t][2]}, f2[3] = {e[t][3], e[t][4], e[t][5]};
      const double p[6] = {f2[0] * f2[0], f2[0] * f2[1], f2[0] * f2[2], f2[1] * f2[1], f2[1] * f2[2], f2[2] * f2[2]};
      const double qq[6] = {f1[0] * f1[0], f1[0] * f1[1], f1[0] * f1[2], f1[1] * f1[1], f1[1] * f1[2], f1[2] * f1[2]};
#pragma unroll
      for (int kl = 0; kl < 6; ++kl)
#pragma unroll
        for (int ac = 0; ac < 6; ++ac) Gl[6 * kl + ac] += p[kl] * qq[ac];
#pragma unroll
      for (int c = 0; c < 3; ++c) ev1[c] += f1[c];
    }
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) Gl[i] = quad_sum(Gl[i]);
#pragma unroll
  for (int c = 0; c < 3; ++c) ev1[c] = quad_sum(ev1[c]);
}

#ifndef PNEC_RANSAC_WAVES_PER_SIMD
#define PNEC_RANSAC_WAVES_PER_SIMD 2
#endif
// ---- helpers of the RANSAC kernels' tails

// InlierExtraction of one pair by one wavefront (the body of select_kernel): eight 64-chunks at a time, first where
// every kept correspondence goes (mask bytes and ballots only), then the copies component by component
__device__ __forceinline__ void compact_pair(int nc, const double *sb, int n, const uint8_t *mk, double *db, int m, int lane) {
  const int sstride = (n + kWave - 1) & ~(kWave - 1), dstride = (m + kWave - 1) & ~(kWave - 1);
  constexpr int kChunks = 8;
  int written = 0;
  for (int base = 0; base < sstride; base += kChunks * kWave) {
    bool in[kChunks];
    int pos[kChunks];
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      const int idx = base + j * kWave + lane;
      in[j] = idx < n && mk[idx] != 0;
      const unsigned long long b = __ballot(in[j]);
      pos[j] = written + __popcll(b & ((1ull << lane) - 1ull));
      written += __popcll(b);
    }
    for (int c = 0; c < nc; ++c) {
      const double *sc = sb + (int64_t)c * sstride + base + lane;
      double *dc = db + (int64_t)c * dstride;
      double v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j) v[j] = in[j] ? sc[j * kWave] : 0.0;
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        if (in[j]) dc[pos[j]] = v[j];
    }
  }
  for (int idx = m + lane; idx < dstride; idx += kWave)  // zero padding of the last 64-chunk
    for (int c = 0; c < nc; ++c) db[(int64_t)c * dstride + idx] = 0.0;
}

// ---- RANSAC kernel shape: ONE QUAD PER HYPOTHESIS, TWO PAIRS PER WAVEFRONT with the hypotheses of both in one queue ----
// A round evaluates up to sixteen hypotheses per pair.  The four lanes of a quad share a hypothesis' minimisation (the
// finite-difference probes and step lengths of one evaluation) and split the correspondences of its sample; everything
// that is one value per pair -- the adaptive bound k, the best count, the iteration counter -- is wave-uniform, so the
// sequential rule of the reference loop is scalar code.  A wavefront owns two pairs: a round prepares up to 32 hypotheses
// (sample, 36 sums, jittered start: per hypothesis in LDS), the sixteen quads PULL them from one queue (es_minimise_queue:
// a quad that has finished a minimisation re-arms on the next hypothesis of the list, whichever pair it belongs to) --
// with one pair per wavefront a round's minimisations run at the pace of the slowest of sixteen and the quads idle half
// of the phase -- and the rest of the round (model, scoring one model after the other with the exact early drop, the
// sequential consume rule) runs per pair with quad j on hypothesis j.  A pair whose partner is done, or that has a
// wavefront to itself (the launch's last wavefronts, a handful of pairs, the per-frame handle's one), takes the second
// slot group for its own next sixteen hypotheses.  A hypothesis' arithmetic does not depend on which quad minimises it,
// when, or which pair shares the wavefront.  (Until round 5 a one-pair-per-wavefront kernel existed beside this one for
// small batches; this kernel with n_double = 0 is that shape, so it went.)
struct Ransac2Lds {
  double Gh[2 * kHypPerRound][36];   // the 36 sums of each hypothesis' sample
  double tv[2 * kHypPerRound][3];    // its start (in) / minimiser (out) in Cayley coordinates
  double te[2 * kHypPerRound][3];    // eigenvector of the smallest eigenvalue at the minimiser (the translation)
  double tev1[2 * kHypPerRound][3];  // sum of the sample's f1 (directional evidence)
  int tsel[2 * kHypPerRound][PNEC_HIP_MAX_RANSAC_SAMPLE];  // the sample
  int tits[2 * kHypPerRound];
  int tlist[2 * kHypPerRound];       // the round's queue: slots of the active hypotheses
  double tkey[2 * kHypPerRound];     // ... and the keys it is ordered by (es_queue_order)
  double models[kHypPerRound][13];   // R (9) + t (3) of the hypotheses of the pair being scored + kModelCapped
  double best_model[2][12];          // R (9) + t (3) per pair
  double G[2][36];                   // sums of the inliers of the best model
  double parked[32];                 // a finished pair waiting for the kernel's end while its slot is lent (see below)
};

// es_minimise_quad<1> for a QUEUE of problems: problem slot s has its sums at Gtab[s], its start at tv[s]; on return
// tv[s] is the minimiser, te[s] the eigenvector there, tits[s] the Newton iterations.  Quad q starts on tlist[q]; a quad
// that is done stores its result and takes the next entry of tlist (in quad order when several finish in one trip).
// One trip of the loop is ONE evaluation for every quad that is busy, exactly the trip of es_minimise_quad -- the same
// arithmetic per problem, hence the same bits.
__device__ __forceinline__ int es_minimise_queue(int n_tasks, const int *tlist, const double (*Gtab)[36], double (*tv)[3],
                                              double (*te)[3], int *tits, double n_scale) {
  enum : int { kInit = 0, kTrial, kShort, kReeval, kDone };
  const int lane = (int)threadIdx.x, quad = lane >> 2, role_of_lane = lane & 3;
  const double h = 1e-6, inv_h = 1.0 / h;
  double v[3] = {0.0, 0.0, 0.0}, eb[3] = {0.0, 0.0, 1.0};
  double f = 0.0, g[3] = {0.0, 0.0, 0.0}, H[9], d[3] = {0.0, 0.0, 0.0};
  double slope = 0.0, alpha = 1.0, trace_cur = 0.0;
  bool damped = false;
  int state = kDone, it = 0, ls = 0, slot = -1;
  bool last_eval = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) H[i] = 0.0;
  auto arm = [&](int s) {
    slot = s;
    v[0] = tv[s][0]; v[1] = tv[s][1]; v[2] = tv[s][2];
    eb[0] = 0.0; eb[1] = 0.0; eb[2] = 1.0;
    f = 0.0; g[0] = g[1] = g[2] = 0.0; d[0] = d[1] = d[2] = 0.0;
    slope = 0.0; alpha = 1.0; trace_cur = 0.0; damped = false;
    it = 0; ls = 0; last_eval = false;
    state = kInit;
  };
  if (quad < n_tasks) arm(tlist[quad]);
  int next = kHypPerRound;  // wave-uniform: the next entry of tlist to hand out
  int trips = 0;            // evaluations as the wavefront executes them (diagnostics)
  [[maybe_unused]] int my_evals = 0;
  for (;;) {
    PNEC_FMARK("q_point");
    ++trips;
#ifdef PNEC_WORK_COUNT
    if (state != kDone) ++my_evals;
#endif
    if (state != kDone) {
      const double *G = Gtab[slot];
      // ---- the point this lane evaluates in this trip
      // (the lane's role goes through an empty asm statement in every trip: left to itself the compiler hoists the
      //  probe offsets and step fractions that depend on it out of the loop, runs out of registers and reloads them
      //  from scratch one by one, each behind its own s_waitcnt -- three trips to memory per evaluation)
      int role = role_of_lane;
      asm volatile("" : "+v"(role));
      double p[3] = {v[0], v[1], v[2]};
      double a_mine = 0.0;
      if (state == kShort) {
        a_mine = alpha * (role == 0 ? 1.0 : (role == 1 ? 0.5 : (role == 2 ? 0.25 : 0.125)));
#pragma unroll
        for (int k = 0; k < 3; ++k) p[k] = v[k] + a_mine * d[k];
      } else {
        if (state == kTrial) {
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = v[k] + d[k];
        }
        p[0] += (role == 1 ? h : 0.0);
        p[1] += (role == 2 ? h : 0.0);
        p[2] += (role == 3 ? h : 0.0);
      }
      double gp[3], Mp[9], ep[3] = {eb[0], eb[1], eb[2]};
      const double fp = es_value_grad<1>(G, p, gp, Mp, ep, state != kInit);
      PNEC_FMARK("q_post");
      const double trace_p = Mp[0] + Mp[4] + Mp[8];
      bool at_new_point = false;
      if (state == kShort) {
        const int pass = (ls + role < 40 && fp <= f + 1e-4 * a_mine * slope + 4e-16 * trace_p) ? 1 : 0;
        const int p0 = quad_broadcast<0>(pass), p1 = quad_broadcast<1>(pass), p2 = quad_broadcast<2>(pass),
                  p3 = quad_broadcast<3>(pass);
        if (p0 | p1 | p2 | p3) {
          alpha *= p0 ? 1.0 : (p1 ? 0.5 : (p2 ? 0.25 : 0.125));
          const double smax = alpha * fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
          for (int k = 0; k < 3; ++k) v[k] = v[k] + alpha * d[k];
          ++it;
          last_eval = smax < 1e-12 || it >= kHypothesisMaxIterations;
          state = kReeval;  // the eigenvector AT the new point is wanted: one more evaluation even at the end
        } else {
          alpha *= 0.0625;
          ls += 4;
          if (ls >= 40) state = kDone;
        }
      } else {
        const double fx = quad_broadcast<0>(fp), trace_x = quad_broadcast<0>(trace_p);
        double gx[3], Hx[9], ex[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          gx[r] = quad_broadcast<0>(gp[r]);
          Hx[3 * r + 0] = (quad_broadcast<1>(gp[r]) - gx[r]) * inv_h;
          Hx[3 * r + 1] = (quad_broadcast<2>(gp[r]) - gx[r]) * inv_h;
          Hx[3 * r + 2] = (quad_broadcast<3>(gp[r]) - gx[r]) * inv_h;
          ex[r] = quad_broadcast<0>(ep[r]);
        }
        Hx[1] = Hx[3] = 0.5 * (Hx[1] + Hx[3]);
        Hx[2] = Hx[6] = 0.5 * (Hx[2] + Hx[6]);
        Hx[5] = Hx[7] = 0.5 * (Hx[5] + Hx[7]);
        bool take = true;
        if (state == kTrial) {
          take = fx <= f + 1e-4 * slope + 4e-16 * trace_x;
          if (!take) {  // the gradient judges the full step when the value cannot (see es_minimise_quad)
            const double gmax_old = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
            const double gmax_new = fmax(fabs(gx[0]), fmax(fabs(gx[1]), fabs(gx[2])));
            take = (fx - f) <= 1e-13 * trace_x && gmax_new < gmax_old;
          }
          if (take) {
            const double smax = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = v[k] + d[k];
            ++it;
            if (smax < (damped ? 1e-12 : kHypothesisStepDone) || it >= kHypothesisMaxIterations) state = kDone;
          } else {
            state = kShort;
            alpha = 0.5;
            ls = 1;
          }
        }
        if (take) {
          f = fx;
          trace_cur = trace_x;
#pragma unroll
          for (int i = 0; i < 3; ++i) { g[i] = gx[i]; eb[i] = ex[i]; }
#pragma unroll
          for (int i = 0; i < 9; ++i) H[i] = Hx[i];
          if (last_eval) state = kDone;
          at_new_point = state != kDone;
        }
      }
      PNEC_FMARK("q_head");
      if (at_new_point) {
        const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        if (gmax <= fmax(1e-14 * (1.0 + fabs(f)) * n_scale, 1.1e-13 * trace_cur)) {
          state = kDone;
        } else {
          PNEC_DBG_WAVE(16);           // queue: iteration heads as the wavefront executes them
          const bool ok = levenberg_direction(H, g, role, d, damped);
          if (ok) {
            slope = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
            state = kTrial;
          } else {
            state = kDone;
          }
        }
      }
    }
    PNEC_FMARK("q_fin");
    // ---- quads that have finished: park the result, take the next problem of the queue (in quad order)
    const bool fin = state == kDone && slot >= 0;
    const int role = role_of_lane;
    if (fin && role == 0) {
      tv[slot][0] = v[0]; tv[slot][1] = v[1]; tv[slot][2] = v[2];
      te[slot][0] = eb[0]; te[slot][1] = eb[1]; te[slot][2] = eb[2];
      tits[slot] = it;
    }
    const unsigned long long fb = __builtin_amdgcn_ballot_w64(fin && role == 0);
    if (fin) {
      const int rank = __builtin_popcountll(fb & ((1ull << (lane & ~3)) - 1ull));
      const int idx = next + rank;
      slot = -1;
      if (idx < n_tasks) arm(tlist[idx]);
    }
    next += __builtin_popcountll(fb);
    if (__builtin_amdgcn_ballot_w64(state != kDone) == 0ull) break;
    PNEC_FMARK("q_loop_end");
  }
  PNEC_FMARK("q_after");
#ifdef PNEC_WORK_COUNT
  if (role_of_lane == 0) PNEC_WORK_ADD(kWkRansacEvals, my_evals);
#endif
  return trips;
}

// SCHEME: which iteration minimises a hypothesis' eigenvalue (0: es_minimise_queue; 1, 2: es_minimise_queue_alt, where
// every hypothesis is scored -- there is no cap that voids a model).
// PHASE (an A/B form of round 5, -DPNEC_RANSAC_TWO_LAUNCH_AB: the stage as TWO launches; the default library only has
// PHASE 0): 0 = a pair's whole RANSAC in this launch; 1 = its FIRST
// round only -- a pair that is done after it (six in ten at 10 % mismatches) is finished here, the others leave where
// their sequential rule stands (RansacArgs::st_*) and their index in a list; 2 = the listed pairs, two per wavefront by a
// grid-stride loop, from that state to the end.  Why: a launch ends with its slowest wavefronts, and in ONE launch the
// pairs that need a second and third round sit wherever the batch has them -- the ones dispatched late end it late (a
// quarter of the launch was that tail; the opt-in launch-order hint cured it with knowledge from an earlier call).  The
// first rounds are all alike (no tail to speak of), and the long pairs start TOGETHER at the top of the second launch.
// A hypothesis' arithmetic does not depend on the launch it runs in: same bits as PHASE 0 -- and measured 8 % slower: the
// first rounds of other pairs are what fills the long pairs' tails in ONE launch (launch_ransac_eigensolver).
template <int SCHEME, int PHASE>
__global__ __launch_bounds__(kWave, PNEC_RANSAC_WAVES_PER_SIMD) void ransac2_eigensolver_kernel(const RansacArgs a) {
  const int lane = threadIdx.x;
  const int hyp = lane >> 2, role = lane & 3;
  __shared__ Ransac2Lds lds;
  const int ss = a.sample_size;  // <= PNEC_HIP_MAX_RANSAC_SAMPLE (checked by the caller)
  auto lds_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // PHASE 2: the listed pairs in the order of the buckets, the one with the most rounds still asked for first
  [[maybe_unused]] int st_n[kStBuckets];
  int64_t n_listed = 0;
  if constexpr (PHASE == 2) {
#pragma unroll
    for (int b = 0; b < kStBuckets; ++b) {
      st_n[b] = a.st_count[b];
      n_listed += st_n[b];
    }
  }
  [[maybe_unused]] auto listed_pair = [&](int64_t e) -> int64_t {   // entry e of the concatenated lists (e < n_listed)
    int64_t left = e;
#pragma unroll
    for (int b = kStBuckets - 1; b >= 0; --b) {
      if (left < st_n[b]) return (int64_t)a.st_list[(int64_t)b * a.n_pairs + left];
      left -= st_n[b];
    }
    return 0;
  };
  for (int64_t work = (int64_t)blockIdx.x; PHASE != 2 || 2 * work < n_listed; work += (int64_t)gridDim.x) {
  // ---- the two pairs' wave-uniform state
  int64_t pair[2];
  int n[2], stride[2], it[2] = {0, 0}, best_count[2] = {-1, -1};
  const double *base[2];
  double k[2] = {1.0, 1.0}, v0[2][3], R0[2][9];
  bool stop[2], can_sample[2], exists[2];
  [[maybe_unused]] bool unfinished[2] = {false, false};   // PHASE 1: the pair goes on in the second launch
  [[maybe_unused]] int rounds_done = 0;
  bool lent = false;  // slot 1 works for slot 0's pair (its second sixteen hypotheses of a round)
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    const int64_t blk = work;
    if constexpr (PHASE == 2) {
      exists[pp] = 2 * blk + pp < n_listed;
      pair[pp] = exists[pp] ? listed_pair(2 * blk + pp) : 0;
    } else {
      pair[pp] = blk < a.n_double ? 2 * blk + pp : a.n_double + blk;   // (= 2 n_double + (blk - n_double) for slot 0)
      exists[pp] = pair[pp] < a.n_pairs && (blk < a.n_double || pp == 0);
      if (a.order && exists[pp]) pair[pp] = a.order[pair[pp]];
    }
    const int64_t pq = exists[pp] ? pair[pp] : 0;
    n[pp] = exists[pp] ? a.count[pq] : 0;
    stride[pp] = (n[pp] + kWave - 1) & ~(kWave - 1);
    base[pp] = a.data + a.block_offset[pq];
    double q0[4] = {a.init_q[4 * pq], a.init_q[4 * pq + 1], a.init_q[4 * pq + 2], a.init_q[4 * pq + 3]};
    const double qn = 1.0 / sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
    for (int c = 0; c < 4; ++c) q0[c] *= qn;
    rot_from_quat(q0, R0[pp]);
    rot_to_cayley(R0[pp], v0[pp]);
    can_sample[pp] = exists[pp] && n[pp] >= ss && ss >= 1;
    stop[pp] = !can_sample[pp];
    if constexpr (PHASE == 2) {   // from where the first launch left the pair's rule
      if (exists[pp]) {
        it[pp] = (int)a.st_it[pq];
        best_count[pp] = (int)a.st_best[pq];
        k[pp] = a.st_k[pq];
        if (lane < 12) lds.best_model[pp][lane] = a.st_model[12 * pq + lane];
      }
    }
  }
  if constexpr (PHASE == 2) lds_sync();
  unsigned long long ph_clk[kPhCount] = {0};
  const unsigned long long ph_start = a.trace ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  PNEC_PHASE_BEGIN();
  for (;;) {
    bool go[2];
    int needed[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      go[pp] = !stop[pp] && (double)it[pp] < k[pp];
      if (!go[pp]) stop[pp] = true;
    }
    if constexpr (PHASE == 1) {
      if (rounds_done == 1) {   // the first round is consumed: who goes on does so in the second launch
        // (a slot that was lent holds no pair of its own: the pair is slot 0's)
        unfinished[0] = go[0];
        unfinished[1] = go[1] && !lent;
        break;
      }
      ++rounds_done;
    }
    // ---- A pair whose partner is done takes the partner's slot as well: from its next round on, slot 1 works on the
    // pair's SECOND sixteen hypotheses of the round (it + 16 ...), consumed after slot 0's by the same sequential rule.
    // The long pairs are the ones a launch ends with (a quarter of it is their tail): half as many rounds for them.
    // The finished pair waits in LDS for the kernel's end.  All of this is state shuffling at a round's boundary --
    // the round's code does not know; and since neither slot nor round enters a hypothesis' arithmetic: same bits.
    if (!lent && go[0] != go[1] && !a.chained) {   // (chained starts: hypothesis it + 16 does not exist before it + 15's model)
      auto park = [&](auto dc) {  // the done (or absent) pair of slot D
        constexpr int D = decltype(dc)::value;
        if (lane == 0) {
          lds.parked[0] = (double)it[D];
          lds.parked[1] = (double)n[D];
          lds.parked[2] = exists[D] ? 1.0 : 0.0;
          lds.parked[3] = can_sample[D] ? 1.0 : 0.0;
          lds.parked[4] = (double)pair[D];   // (exact: pair indices are far below 2^53)
#pragma unroll
          for (int i = 0; i < 9; ++i) lds.parked[5 + i] = R0[D][i];
        }
        if (lane < 12) lds.parked[16 + lane] = lds.best_model[D][lane];
      };
      if (go[0]) {
        park(std::integral_constant<int, 1>{});
      } else {
        park(std::integral_constant<int, 0>{});
        lds_sync();
        // the pair that goes on moves to slot 0 (slot 1's hypotheses are consumed second)
        pair[0] = pair[1]; n[0] = n[1]; stride[0] = stride[1]; base[0] = base[1];
        it[0] = it[1]; best_count[0] = best_count[1]; k[0] = k[1];
        can_sample[0] = can_sample[1]; exists[0] = exists[1]; stop[0] = false; go[0] = true;
#pragma unroll
        for (int i = 0; i < 3; ++i) v0[0][i] = v0[1][i];
#pragma unroll
        for (int i = 0; i < 9; ++i) R0[0][i] = R0[1][i];
        if (lane < 12) lds.best_model[0][lane] = lds.best_model[1][lane];
      }
      pair[1] = pair[0]; n[1] = n[0]; stride[1] = stride[0]; base[1] = base[0];
      can_sample[1] = can_sample[0];
#pragma unroll
      for (int i = 0; i < 3; ++i) v0[1][i] = v0[0][i];
      lent = true;
      lds_sync();
    }
    if (lent) {  // slot 1 continues where slot 0's sixteen end
      it[1] = it[0] + kHypPerRound;
      k[1] = k[0];
      best_count[1] = best_count[0];
      stop[1] = false;
      go[1] = go[0] && it[0] > 0 && (double)it[1] < k[1];  // (the first round evaluates 16 hypotheses, as ever)
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      // the first round evaluates all 16 hypotheses; a later round only those that can still be consumed (see the
      // one-pair kernel)
      needed[pp] = !go[pp] ? 0 : (it[pp] == 0 ? kHypPerRound : (int)fmin(ceil(k[pp] - (double)it[pp]), (double)kHypPerRound));
      if (a.chained && needed[pp] > 1) needed[pp] = 1;   // the next hypothesis' start is this one's model
    }
    if (!go[0] && !go[1]) break;  // wave-uniform
    // ---- prepare the round's hypotheses: quad j samples hypothesis j of each pair that goes on.
    // The draws of BOTH slots in one pass: lanes 0, 1 of a quad draw slot 0's sample and jitter, lanes 2, 3 slot 1's,
    // and the quad gets each by broadcast (until round 4 all four lanes drew slot 0's, then all four slot 1's: the
    // generator -- a 64-bit hash per draw -- and the draw-until-distinct loop were a seventh of the kernel's time).
    // Same draws, same samples.
    int smp_mine[PNEC_HIP_MAX_RANSAC_SAMPLE];
    {
      const bool second = role >= 2;
      const bool act_mine = hyp < (second ? needed[1] : needed[0]);
      const unsigned long long hh_m = (unsigned long long)((second ? it[1] : it[0]) + hyp);
      const unsigned long long pid_m = a.pair_id_base + (unsigned long long)(second ? pair[1] : pair[0]);
#pragma unroll
      for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) smp_mine[j] = -1;
      if (act_mine) ransac_sample_regs(a.seed, pid_m, hh_m, second ? n[1] : n[0], ss, smp_mine);
    }
    // the start's jitter: six hashes on four lanes in two turns
    double jit[2][3];
    {
      const unsigned long long pid0 = a.pair_id_base + (unsigned long long)pair[0], pid1 = a.pair_id_base + (unsigned long long)pair[1];
      const unsigned long long hh0 = (unsigned long long)(it[0] + hyp), hh1 = (unsigned long long)(it[1] + hyp);
      // turn A: lanes 0..2 -> slot 0's components 0..2, lane 3 -> slot 1's component 0; turn B: lanes 0, 1 -> slot 1's 1, 2
      const double ja = rng_uniform(a.seed, role < 3 ? pid0 : pid1, role < 3 ? hh0 : hh1, 1000ull + (unsigned long long)(role < 3 ? role : 0));
      const double jb = rng_uniform(a.seed, pid1, hh1, 1001ull + (unsigned long long)(role & 1));
      jit[0][0] = quad_broadcast<0>(ja); jit[0][1] = quad_broadcast<1>(ja); jit[0][2] = quad_broadcast<2>(ja);
      jit[1][0] = quad_broadcast<3>(ja); jit[1][1] = quad_broadcast<0>(jb); jit[1][2] = quad_broadcast<1>(jb);
    }
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const int slot = kHypPerRound * pp + hyp;
      const bool active = hyp < needed[pp];
      // the sample in registers, its gathers all in flight together (ransac_sample_regs / ransac_sample_sums above); the
      // model phase reads the sample from LDS
      int smp[PNEC_HIP_MAX_RANSAC_SAMPLE];
#pragma unroll
      for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j)
        smp[j] = pp == 0 ? quad_broadcast<0>(smp_mine[j]) : quad_broadcast<2>(smp_mine[j]);
      if (active && role == 0) PNEC_WORK_ADD(kWkRansacHyps, 1);
      double ev1[3], Gl[36];
      ransac_sample_sums(base[pp], stride[pp], ss, active, smp, role, Gl, ev1);
      if (role == 0 && active) {
#pragma unroll
        for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) lds.tsel[slot][j] = smp[j];
#pragma unroll
        for (int i = 0; i < 36; ++i) lds.Gh[slot][i] = Gl[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          lds.tev1[slot][c] = ev1[c];
          lds.tv[slot][c] = v0[pp][c] + (jit[pp][c] - 0.5) * 2.0 * 0.01;
        }
      }
    }
    // the round's queue: pair 0's active hypotheses, then pair 1's
    const int n_tasks = needed[0] + needed[1];
    if (lane < 2 * kHypPerRound) {
      const int pp = lane >= needed[0] ? 1 : 0;
      const int j = pp ? lane - needed[0] : lane;
      if (lane < n_tasks) lds.tlist[lane] = kHypPerRound * pp + j;
    }
    lds_sync();
    // ---- the queue's order: likely-long minimisations first (es_queue_order).  The sixteen quads start on the first
    // sixteen entries together and take the others as they finish; a round is as long as its busiest quad, and a long
    // minimisation handed out late ends it (replayed from the checker's trip counts, tools/sim_ransac_queue.py: 25.7
    // trips for a round of 32 in the order of the slots, 20.7 with the lengths known).  What a minimisation will take is
    // not known, but how far its start is from a rank-two M says much of it: a clean sample starts near its minimum, a
    // contaminated one does not, and lambda_1 / lambda_2 of M at the start -- from M's invariants alone,
    // det tr / c2^2, no eigenpair -- ranks the lengths with a correlation of 0.69 (22.1 trips).  One lane per task
    // composes its M (a third of an evaluation, once per round) and ranks its key among the others'.  Scheduling only:
    // a hypothesis' arithmetic does not depend on when or where it is minimised.
    if (n_tasks > kHypPerRound) {  // (wave-uniform; with sixteen or fewer, all start together)
      double key = 0.0;
      int slot_l = 0;
      if (lane < n_tasks) {
        slot_l = lds.tlist[lane];
        double Gr[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) Gr[i] = lds.Gh[slot_l][i];
        const double vv[3] = {lds.tv[slot_l][0], lds.tv[slot_l][1], lds.tv[slot_l][2]};
        double Rk[9], Mk[9];
        cayley_to_rot(vv, Rk);
        sums_to_m(Gr, Rk, Mk);
        const double m01 = 0.5 * (Mk[1] + Mk[3]), m02 = 0.5 * (Mk[2] + Mk[6]), m12 = 0.5 * (Mk[5] + Mk[7]);
        const double k00 = Mk[4] * Mk[8] - m12 * m12, k11 = Mk[0] * Mk[8] - m02 * m02, k22 = Mk[0] * Mk[4] - m01 * m01;
        const double c2 = k00 + k11 + k22;
        const double det = Mk[0] * k00 - m01 * (m01 * Mk[8] - m12 * m02) + m02 * (m01 * m12 - Mk[4] * m02);
        key = det * (Mk[0] + Mk[4] + Mk[8]) * fast_rcp(c2 * c2);
        if (!(key > 0.0) || !finite_d(key)) key = 0.0;
        lds.tkey[lane] = key;
      }
      lds_sync();
      if (lane < n_tasks) {
        int rank = 0;
        for (int j = 0; j < n_tasks; ++j) {
          const double kj = lds.tkey[j];
          rank += (kj > key || (kj == key && j < lane)) ? 1 : 0;
        }
        lds.tlist[rank] = slot_l;
      }
      lds_sync();
    }
    PNEC_PHASE_END(kRpSample);
    int trips;
    if constexpr (SCHEME == 0) trips = es_minimise_queue(n_tasks, lds.tlist, lds.Gh, lds.tv, lds.te, lds.tits, (double)ss);
    else trips = es_minimise_queue_alt<SCHEME>(n_tasks, lds.tlist, lds.Gh, lds.tv, lds.te, lds.tits, nullptr);
    lds_sync();
    PNEC_PHASE_END(kRpNewton);
    if (a.trace) {  // diagnostics (per wavefront = two pairs; halved on the way out): Newton iterations of the round's
                    // hypotheses (sum), evaluations as executed (in the slot the one-pair kernel uses for the slowest
                    // quad's iterations), rounds
      int its = (lane < 2 * kHypPerRound && ((lane < kHypPerRound) ? lane < needed[0] : lane - kHypPerRound < needed[1]))
                    ? lds.tits[lane] : 0;
      ph_clk[8] += 2ull * (unsigned long long)wave_allreduce_sum((double)its);
      ph_clk[9] += 2ull * (unsigned long long)trips;
      ph_clk[10] += 2;
    }
    // ---- per pair: model of hypothesis `hyp`, scoring, the sequential rule
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      if (!go[pp]) continue;  // wave-uniform
      const int slot = kHypPerRound * pp + hyp;
      const bool active = hyp < needed[pp];
      const double *bs = base[pp];
      const int st = stride[pp], nn = n[pp];
      double v[3] = {0, 0, 0}, R[9], t[3] = {0, 0, 1};
      if (active) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { v[c] = lds.tv[slot][c]; t[c] = lds.te[slot][c]; }
      }
      cayley_to_rot(v, R);
      {
        // directional evidence sum t.(f1 - R f2) over the sample
        double ev = 0.0;
        for (int j = role; j < (active ? ss : 0); j += 4) {
          const int idx = lds.tsel[slot][j];
          const double f2[3] = {bs[(int64_t)3 * st + idx], bs[(int64_t)4 * st + idx], bs[(int64_t)5 * st + idx]};
          const double u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                               R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
          ev -= t[0] * u[0] + t[1] * u[1] + t[2] * u[2];
        }
        const double e1[3] = {active ? lds.tev1[slot][0] : 0.0, active ? lds.tev1[slot][1] : 0.0, active ? lds.tev1[slot][2] : 0.0};
        ev = (t[0] * e1[0] + t[1] * e1[1] + t[2] * e1[2]) + quad_sum(ev);
        if (ev < 0.0) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
      }
      PNEC_PHASE_END(kRpModel);
      // the models go to LDS; then one hypothesis after the other, in the order of the sequential rule, is scored by
      // the whole wavefront and consumed (see the one-pair kernel)
      if (active && role == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) lds.models[hyp][i] = R[i];
        lds.models[hyp][9] = t[0]; lds.models[hyp][10] = t[1]; lds.models[hyp][11] = t[2];
        lds.models[hyp][kModelCapped] = (SCHEME == 0 && lds.tits[slot] >= kHypothesisMaxIterations) ? 1.0 : 0.0;
      }
      lds_sync();
      ScoreTiles tiles;  // the pair's bearings for the scoring (issue and wait back to back, see the one-pair kernel)
      score_tiles_load(tiles, bs, st, nn, lane);
      int winner = -1;
      for (int j = 0; j < needed[pp]; ++j) {
        if (!((double)it[pp] < k[pp])) { stop[pp] = true; break; }
        double Rj[9], tj[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rj[i] = lds.models[j][i];
        tj[0] = lds.models[j][9]; tj[1] = lds.models[j][10]; tj[2] = lds.models[j][11];
        const int cj = lds.models[j][kModelCapped] != 0.0 ? 0 : model_inliers_until_beaten(tiles, bs, st, nn, Rj, tj, a.threshold, lane, best_count[pp]);
        if (cj > best_count[pp]) {
          best_count[pp] = cj;
          winner = j;
          const double w = (double)cj / (double)nn;
          double p_no = 1.0 - pow_sample(w, ss);
          p_no = fmax(2.220446049250313e-16, p_no);
          p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
          k[pp] = log(1.0 - 0.99) / log(p_no);
        }
        ++it[pp];
        if (it[pp] > a.max_iterations) { stop[pp] = true; break; }
      }
      PNEC_PHASE_END(kRpScore);
      if (a.chained && needed[pp] > 0 && lds.models[0][kModelCapped] == 0.0) {
        // the model just scored stays "in the adapter": the next hypothesis starts from its rotation (wave-uniform)
        double Rl[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rl[i] = lds.models[0][i];
        rot_to_cayley(Rl, v0[pp]);
      }
      if (winner >= 0 && lane < 12) lds.best_model[pp][lane] = lds.models[winner][lane];
      lds_sync();
      if (lent) {
        if (pp == 0) {  // slot 1 consumes next, from where this slot's rule stands now
          k[1] = k[0];
          best_count[1] = best_count[0];
          if (stop[0]) go[1] = false;
        } else {        // what slot 1 consumed is the pair's: hand it back
          // (slot 1 only runs when slot 0 consumed its full sixteen: it[0] is where slot 1 started)
          it[0] = it[1];
          k[0] = k[1];
          best_count[0] = best_count[1];
          stop[0] = stop[0] || stop[1];
          if (winner >= 0 && lane < 12) lds.best_model[0][lane] = lds.best_model[1][lane];
          lds_sync();
        }
      }
      PNEC_PHASE_END(kRpConsume);
    }
  }
  if (lent) {  // the pair that waited goes back into slot 1 for the common ending
    lds_sync();
    it[1] = (int)lds.parked[0];
    n[1] = (int)lds.parked[1];
    exists[1] = lds.parked[2] != 0.0;
    can_sample[1] = lds.parked[3] != 0.0;
    pair[1] = (int64_t)lds.parked[4];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[1][i] = lds.parked[5 + i];
    stride[1] = (n[1] + kWave - 1) & ~(kWave - 1);
    base[1] = a.data + a.block_offset[exists[1] ? pair[1] : 0];
    if (lane < 12) lds.best_model[1][lane] = lds.parked[16 + lane];
    lds_sync();
  }
  // ---- per pair: inliers of the best model (all correspondences when sampling is impossible), their 36 sums, the
  // first inlier; handed to es_batch_kernel<kEpiTranslation> as in the one-pair kernel
#pragma unroll
  for (int pp = 0; pp < 2; ++pp) {
    if (!exists[pp]) continue;
    if constexpr (PHASE == 1) {
      if (unfinished[pp]) {   // leave the rule's state and the pair's index for the second launch
        const int64_t pq = pair[pp];
        if (lane < 12) a.st_model[12 * pq + lane] = lds.best_model[pp][lane];
        if (lane == 0) {
          a.st_it[pq] = (double)it[pp];
          a.st_best[pq] = (double)best_count[pp];
          a.st_k[pq] = k[pp];
          // (the bucket: how many more rounds of sixteen the rule asks for as it stands -- k only falls from here on)
          const double more = ceil((k[pp] - (double)it[pp]) * (1.0 / kHypPerRound));
          const int bucket = more >= (double)a.st_buckets ? a.st_buckets - 1 : (more <= 1.0 ? 0 : (int)more - 1);
          a.st_list[(int64_t)bucket * a.n_pairs + atomicAdd(a.st_count + bucket, 1)] = (int32_t)pq;
        }
        continue;
      }
    }
    const double *bs = base[pp];
    const int st = stride[pp], nn = n[pp];
    double bR[9], bt[3] = {0.0, 0.0, 1.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) bR[i] = R0[pp][i];
    if (can_sample[pp]) {
#pragma unroll
      for (int i = 0; i < 9; ++i) bR[i] = lds.best_model[pp][i];
      bt[0] = lds.best_model[pp][9]; bt[1] = lds.best_model[pp][10]; bt[2] = lds.best_model[pp][11];
    }
    double acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    int my_count = 0, my_first = 0x7fffffff;
    const int64_t aos0 = a.offsets[pair[pp]];
    // (the loop reads every 64 correspondences where it uses them.  Measured and taken back in round 6: the first 512 from
    // ONE batch of 48 loads, as the scoring does, and InlierExtraction six component planes at a time -- 204 against 202 us
    // per frame, 2.70 against 2.66 ms per 20 000 pairs: the loads were not what this pass waits for, and the 96 registers of
    // the batch cost fifteen more spilled ones)
    for (int idx = lane; idx < nn; idx += kWave) {
      const double f1[3] = {bs[idx], bs[(int64_t)st + idx], bs[(int64_t)2 * st + idx]};
      const double f2[3] = {bs[(int64_t)3 * st + idx], bs[(int64_t)4 * st + idx], bs[(int64_t)5 * st + idx]};
      bool in = true;
      if (can_sample[pp]) in = reprojection_score(f1, f2, bR, bt) < a.threshold;
      if (a.out_mask) a.out_mask[aos0 + idx] = in ? 1 : 0;
      if (in) {
        ++my_count;
        if (idx < my_first) my_first = idx;
        const double p[6] = {f2[0] * f2[0], f2[0] * f2[1], f2[0] * f2[2], f2[1] * f2[1], f2[1] * f2[2], f2[2] * f2[2]};
        const double qq[6] = {f1[0] * f1[0], f1[0] * f1[1], f1[0] * f1[2], f1[1] * f1[1], f1[1] * f1[2], f1[2] * f1[2]};
#pragma unroll
        for (int kl = 0; kl < 6; ++kl)
#pragma unroll
          for (int ac = 0; ac < 6; ++ac) acc[6 * kl + ac] = __builtin_fma(p[kl], qq[ac], acc[6 * kl + ac]);
      }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      const double sres = wave_allreduce_sum(acc[i]);
      if (lane == 0) lds.G[pp][i] = sres;
    }
    const int total = (int)wave_allreduce_sum((double)my_count);
    if (lane == 0) PNEC_WORK_ADD(kWkRansacInlierCorr, nn);
    int first = my_first;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(first, off);
      first = o < first ? o : first;
    }
    lds_sync();
    PNEC_PHASE_END(kRpInliers);
    if (lane < 36) a.scratch.G[36 * pair[pp] + lane] = lds.G[pp][lane];
    if (lane == 0) {
      double vv[3];
      rot_to_cayley(bR, vv);
      a.scratch.v0[3 * pair[pp]] = vv[0]; a.scratch.v0[3 * pair[pp] + 1] = vv[1]; a.scratch.v0[3 * pair[pp] + 2] = vv[2];
      a.scratch.n_scale[pair[pp]] = (double)(total > 0 ? total : 1);
      a.scratch.first[pair[pp]] = total > 0 ? first : -1;
      if (a.out_count) a.out_count[pair[pp]] = total;
      if (a.out_iterations) a.out_iterations[pair[pp]] = it[pp];
      if (a.sel_data) a.sel_count[pair[pp]] = total;
      if (a.trace) {
        ph_clk[kRpTotal] = (__builtin_amdgcn_s_memtime() - ph_start) / 2;  // two pairs shared this wavefront
        for (int kk = 0; kk < kPhCount; ++kk) a.trace[kPhCount * pair[pp] + kk] = kk == kRpTotal ? ph_clk[kk] : ph_clk[kk] / 2;
        // the wavefront's place on the launch's timeline, in the constant-rate counter all CUs share (10 ns ticks;
        // in the slots of the one-pair form's counters)
        a.trace[kPhCount * pair[pp] + kRpFinal] = rt_start;
        a.trace[kPhCount * pair[pp] + 11] = __builtin_amdgcn_s_memrealtime();
      }
    }
    if (a.sel_data)  // InlierExtraction (this wavefront's own mask bytes back: each lane reads what it wrote)
      compact_pair(a.nc, bs, nn, a.out_mask + aos0, a.sel_data + a.sel_block[pair[pp]], total, lane);
  }
  if constexpr (PHASE != 2) break;
  lds_sync();   // (the next two pairs of the list reuse the LDS)
  }
}

// ---- InlierExtraction (pnec.cc:210-229): compact the masked correspondences of every pair into a
// new batch, order preserved.  One wavefront per pair -- or, with gridDim.y = nc, per pair and component plane (a
// handful of pairs, the per-frame handle's one: twelve wavefronts copy where one did); positions by ballot prefix
// counts.  dst_count_in: the inliers per pair (from the mask: mask_count_kernel, or RANSAC's own count of the mask it
// wrote); when dst_count_out differs it receives a copy, and single_offsets (a batch of ONE pair) the new AoS offsets.
__global__ __launch_bounds__(kWave) void select_kernel(int nc, const double *src, const int64_t *src_block,
                                                       const int64_t *src_offsets, const int32_t *src_count,
                                                       const uint8_t *mask, double *dst,
                                                       const int64_t *dst_block, const int32_t *dst_count_in,
                                                       int32_t *dst_count_out, int64_t *single_offsets) {
  const int64_t pair = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = src_count[pair], m = dst_count_in[pair];
  const int c_begin = gridDim.y > 1 ? (int)blockIdx.y : 0, c_end = gridDim.y > 1 ? (int)blockIdx.y + 1 : nc;
  if (blockIdx.y == 0 && lane == 0) {
    if (dst_count_out != dst_count_in) dst_count_out[pair] = m;
    if (single_offsets) {
      single_offsets[0] = 0;
      single_offsets[1] = m;
    }
  }
  const int sstride = (n + kWave - 1) & ~(kWave - 1), dstride = (m + kWave - 1) & ~(kWave - 1);
  const double *sb = src + src_block[pair];
  double *db = dst + dst_block[pair];
  const uint8_t *mk = mask + src_offsets[pair];
  // eight 64-chunks at a time: first where every kept correspondence goes (mask bytes and ballots only), then
  // the copies component by component with all eight loads of a component in flight together
  constexpr int kChunks = 8;
  int written = 0;
  for (int base = 0; base < sstride; base += kChunks * kWave) {
    bool in[kChunks];
    int pos[kChunks];
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      const int idx = base + j * kWave + lane;
      in[j] = idx < n && mk[idx] != 0;
      const unsigned long long b = __ballot(in[j]);
      pos[j] = written + __popcll(b & ((1ull << lane) - 1ull));
      written += __popcll(b);
    }
    for (int c = c_begin; c < c_end; ++c) {
      const double *sc = sb + (int64_t)c * sstride + base + lane;
      double *dc = db + (int64_t)c * dstride;
      double v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j) v[j] = in[j] ? sc[j * kWave] : 0.0;
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        if (in[j]) dc[pos[j]] = v[j];
    }
  }
  for (int idx = m + lane; idx < dstride; idx += kWave)  // zero padding of the last 64-chunk
    for (int c = c_begin; c < c_end; ++c) db[(int64_t)c * dstride + idx] = 0.0;
}

// the eigensolver on a pair's sums with PNEC::Eigensolver's tail, by scheme
static void launch_es_batch_translation(const EsBatchArgs &b, int scheme, hipStream_t stream) {
  const dim3 grid(es_batch_blocks(b.n_pairs)), block(kWave);
  if (scheme == 1) hipLaunchKernelGGL((es_batch_alt_kernel<kEpiTranslation, 1>), grid, block, 0, stream, b, 1);
  else if (scheme == 2) hipLaunchKernelGGL((es_batch_alt_kernel<kEpiTranslation, 2>), grid, block, 0, stream, b, 1);
  else hipLaunchKernelGGL(es_batch_kernel<kEpiTranslation>, grid, block, 0, stream, b, 1);
}

hipError_t launch_select(int nc, const double *src, const int64_t *src_block, const int64_t *src_offsets,
                         const int32_t *src_count, const uint8_t *mask, double *dst, const int64_t *dst_block,
                         const int32_t *dst_count_in, int32_t *dst_count_out, int64_t *single_offsets, int64_t n_pairs,
                         hipStream_t stream) {
  // a wavefront per component plane while the pairs alone would not fill the GPU (PNEC_SELECT_SPLIT=0|1 forces: A/B)
  static const int forced = [] {
    const char *ev = std::getenv("PNEC_SELECT_SPLIT");
    return ev && *ev ? std::atoi(ev) : -1;
  }();
  const bool split = forced >= 0 ? forced != 0 : n_pairs < 4096;
  hipLaunchKernelGGL(select_kernel, dim3((unsigned)n_pairs, split ? (unsigned)nc : 1u), dim3(kWave), 0, stream, nc, src,
                     src_block, src_offsets, src_count, mask, dst, dst_block, dst_count_in, dst_count_out, single_offsets);
  return hipGetLastError();
}

// sel (optional): InlierExtraction fused into each pair's last pass -- nc component planes, target planes / block layout /
// counts (+ the AoS offsets of a ONE-pair batch).
// Launch order of the two-pair form from the iteration counts of an EARLIER solve of the same pairs (or of their
// predecessors in a stream of frames): a launch ends with its slowest wavefronts, and the pairs that need a second and
// third round of hypotheses (a tenth of them; their wavefronts run two to three times the median) end it late when they
// are dispatched late.  Pairs whose earlier count exceeded one round go first, each sharing its wavefront with a pair
// that did not (the finished partner's slot group is lent to the long pair); the rest follow in batch order.  Any
// content of `prev` gives a permutation; results do not depend on the order (draws belong to the pair's global index).
__global__ __launch_bounds__(1024) void ransac_order_kernel(const int32_t *prev, int64_t n, int32_t *order) {
  // Two passes over segments of 32 x 1024 pairs, each thread up to 32 consecutive pairs whose counts are ALL requested
  // before the first is used: (1) how many long pairs there are; (2) every pair's place.  The 1024 per-thread counts are
  // scanned by shuffles inside the sixteen wavefronts + one scan of their totals.  (Until round 6: one dependent load
  // after the other per thread and thread 0 walking the 1024 counts alone, ~40 us.)
  constexpr int kPer = 32;
  __shared__ int wave_tot[17];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto scan = [&](int v, int &excl, int &total) {
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    __syncthreads();
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    if (wave == 0) {
      const int w = lane < 16 ? wave_tot[lane] : 0;
      int winc = w;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(winc, d, 64);
        if (lane >= d) winc += o;
      }
      if (lane < 16) wave_tot[lane] = winc - w;
      if (lane == 15) wave_tot[16] = winc;
    }
    __syncthreads();
    excl = wave_tot[wave] + inc - v;
    total = wave_tot[16];
  };
  int64_t L = 0;
  for (int64_t seg = 0; seg < n; seg += (int64_t)kPer * 1024) {
    const int64_t m = n - seg < (int64_t)kPer * 1024 ? n - seg : (int64_t)kPer * 1024;
    const int64_t per = (m + 1023) / 1024, lo = seg + (per * tid < m ? per * tid : m), hi = seg + (per * (tid + 1) < m ? per * (tid + 1) : m);
    int nl = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) nl += (lo + k < hi && prev[lo + k] > kHypPerRound) ? 1 : 0;
    int excl, total;
    scan(nl, excl, total);
    L += total;
  }
  const int64_t R = n - L, paired = L < R ? L : R;
  int64_t long_before = 0;
  for (int64_t seg = 0; seg < n; seg += (int64_t)kPer * 1024) {
    const int64_t m = n - seg < (int64_t)kPer * 1024 ? n - seg : (int64_t)kPer * 1024;
    const int64_t per = (m + 1023) / 1024, lo = seg + (per * tid < m ? per * tid : m), hi = seg + (per * (tid + 1) < m ? per * (tid + 1) : m);
    bool is_long[kPer];
    int nl = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      is_long[k] = lo + k < hi && prev[lo + k] > kHypPerRound;
      nl += is_long[k] ? 1 : 0;
    }
    int excl, total;
    scan(nl, excl, total);
    int64_t il = long_before + excl, ir = lo - il;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (lo + k < hi) {
        if (is_long[k]) {
          const int64_t pos = il < paired ? 2 * il : paired + il;   // (more long pairs than others: the surplus after the mixed wavefronts)
          order[pos] = (int32_t)(lo + k); ++il;
        } else {
          const int64_t pos = ir < paired ? 2 * ir + 1 : paired + ir;
          order[pos] = (int32_t)(lo + k); ++ir;
        }
      }
    }
    long_before += total;
  }
}

hipError_t launch_ransac_order(const int32_t *iterations, int64_t n_pairs, int32_t *order, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(ransac_order_kernel, dim3(1), dim3(1024), 0, stream, iterations, n_pairs, order);
  return hipGetLastError();
}

// Two-pair form: how many pairs at the END of the launch order get a wavefront of their own.  A launch ends with the
// wavefronts that started last; two-pair wavefronts run ~250 us, one-pair ones about two thirds of that, and while the launch drains the
// slots the shorter ones leave are idle anyway (their worse packing -- 16 minimisations on 16 quads -- costs nothing
// there).  PNEC_RANSAC_TAIL_SINGLES overrides (A/B runs; 0 = every wavefront takes two pairs).  Scheduling only: a
// pair's arithmetic does not depend on the wavefront it shares.
static int64_t ransac_tail_singles(int64_t n_pairs) {
  static const int64_t forced = [] {
    const char *ev = std::getenv("PNEC_RANSAC_TAIL_SINGLES");
    return ev && *ev ? (int64_t)std::atoll(ev) : (int64_t)-1;
  }();
  if (forced >= 0) return forced;
  // measured at 20 000 pairs x 512 (one box, round 4): 0 -> 2.05, 1 024 -> 2.00, 2 048 -> 2.04, 4 096 -> 2.15 ms: about half
  // a generation of the 2 048 wavefront slots, and no more than a sixteenth of a smaller batch
  return std::min<int64_t>(1024, n_pairs / 16);
}

hipError_t launch_ransac_eigensolver(const double *data, const int64_t *block_offset, const int64_t *offsets,
                                     const int32_t *count, int64_t n_pairs, const double *init_q,
                                     unsigned long long seed, unsigned long long pair_id_base, int max_iterations,
                                     int sample_size, double threshold, double *out_q, double *out_t, uint8_t *out_mask,
                                     int32_t *out_count, int32_t *out_iterations, double *scratch_d,
                                     int32_t *scratch_i, hipStream_t stream, hipStream_t tail_stream,
                                     hipEvent_t tail_fork, hipEvent_t tail_done, int sel_nc, double *sel_data,
                                     const int64_t *sel_block, int32_t *sel_count, int64_t *sel_single_offsets,
                                     const int32_t *order, int scheme, int ransac_flags) {
  // ransac_flags: PNEC_HIP_RANSAC_CHAINED_STARTS (include/pnec_hip.h)
  // order (optional): the pairs in launch order (ransac_order_kernel; honoured by the two-pair form)
  // tail_stream (optional): the eigensolver on the inliers (es_batch_kernel, which writes out_q / out_t) runs
  // there, forked from `stream` after the RANSAC kernel -- the caller goes on with work that only needs the
  // masks (InlierExtraction) and makes `stream` wait for tail_done before anything reads out_q / out_t
  if (n_pairs <= 0) return hipSuccess;
  RansacArgs a;
  std::memset(&a, 0, sizeof(a));
  a.scratch = front_scratch(scratch_d, scratch_i, n_pairs);
  a.data = data;
  a.block_offset = block_offset;
  a.offsets = offsets;
  a.count = count;
  a.init_q = init_q;
  a.out_q = out_q;
  a.out_t = out_t;
  a.out_mask = out_mask;
  a.out_count = out_count;
  a.out_iterations = out_iterations;
  a.seed = seed;
  a.pair_id_base = pair_id_base;
  a.n_pairs = n_pairs;
  a.max_iterations = max_iterations;
  a.sample_size = sample_size;
  a.threshold = threshold;
  a.nc = sel_nc;
  a.sel_data = sel_data;
  a.sel_block = sel_block;
  a.sel_count = sel_count;
  a.sel_single_offsets = sel_single_offsets;
  a.chained = (ransac_flags & PNEC_HIP_RANSAC_CHAINED_STARTS) ? 1 : 0;
  const char *tr = std::getenv("PNEC_HIP_TRACE_FRONT");
  if (tr && *tr) {
    const hipError_t e0 = hipMalloc(&a.trace, sizeof(unsigned long long) * kPhCount * (size_t)n_pairs);
    if (e0 != hipSuccess) return e0;
  }
#ifdef PNEC_FRONT_DEBUG
  {
    const unsigned long long zeros[24] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), zeros, sizeof(zeros));
  }
#endif
  // Two pairs per wavefront once there are enough pairs to fill the GPU either way; a handful of pairs (the per-frame
  // handle: one) keep a wavefront each -- half the latency.  Same results, bit for bit.  PNEC_RANSAC_FORM=1|2 forces one /
  // two pairs per wavefront (A/B runs).
  static const int forced_form = [] {
    const char *ev = std::getenv("PNEC_RANSAC_FORM");
    return ev && *ev ? std::atoi(ev) : 0;
  }();
  const bool two = forced_form == 1 ? false : (forced_form == 2 ? true : n_pairs >= 4096);
  a.order = two ? order : nullptr;

  // The stage as TWO launches (first rounds, then the pairs that go on: the kernel's PHASE 1 / 2) is an A/B form, built only
  // with -DPNEC_RANSAC_TWO_LAUNCH_AB and then switched on by PNEC_RANSAC_LAUNCHES=2: bit-identical and, measured in round 5,
  // 8 % SLOWER at 10 % mismatches (1.72 -> 1.86 ms per 20 000 pairs; 14.9 -> 15.9 ms at 30 %) -- in one launch the slots
  // that finished pairs free go to other pairs' first rounds at once, which is what fills the long pairs' tails; two
  // launches put a barrier there (NOTES/round-5.md).
#ifdef PNEC_RANSAC_TWO_LAUNCH_AB
  static const int forced_launches = [] {
    const char *ev = std::getenv("PNEC_RANSAC_LAUNCHES");
    return ev && *ev ? std::atoi(ev) : 0;
  }();
  const bool split = !a.trace && forced_launches == 2;
  static const int forced_buckets = [] {
    const char *ev = std::getenv("PNEC_RANSAC_BUCKETS");
    return ev && *ev ? std::max(1, std::min(kStBuckets, std::atoi(ev))) : kStBuckets;
  }();
  a.st_buckets = forced_buckets;
#else
  constexpr bool split = false;
#endif
  // (two launches: the first rounds are all alike -- no tail to shorten with one-pair wavefronts)
  const int64_t singles = !two ? n_pairs : (split ? 0 : std::min<int64_t>(ransac_tail_singles(n_pairs), n_pairs) & ~(int64_t)1);  // (even: the rest pairs up)
  a.n_double = (n_pairs - singles + 1) / 2;
  const int64_t blocks = a.n_double + (n_pairs - std::min<int64_t>(2 * a.n_double, n_pairs));
  a.st_k = a.scratch.st_k; a.st_it = a.scratch.st_it; a.st_best = a.scratch.st_best; a.st_model = a.scratch.st_model;
  a.st_list = a.scratch.st_list; a.st_count = a.scratch.st_count;
  const dim3 g1((unsigned)blocks), bl(kWave);
#ifdef PNEC_RANSAC_TWO_LAUNCH_AB
  // (the second launch: as many blocks as there could be listed pairs -- the hardware hands them out as slots free up; a
  //  block beyond the list's end leaves at once.  A fixed grid looping over the list would tie long pairs to blocks.)
  const dim3 g2((unsigned)std::max<int64_t>(1, (n_pairs + 1) / 2));
#define PNEC_RANSAC_LAUNCH(S)                                                                                 \
  if (!split) {                                                                                               \
    hipLaunchKernelGGL((ransac2_eigensolver_kernel<S, 0>), g1, bl, 0, stream, a);                             \
  } else {                                                                                                    \
    hipError_t em = hipMemsetAsync(a.st_count, 0, sizeof(int32_t) * kStBuckets, stream);                      \
    if (em != hipSuccess) return em;                                                                          \
    hipLaunchKernelGGL((ransac2_eigensolver_kernel<S, 1>), g1, bl, 0, stream, a);                             \
    hipLaunchKernelGGL((ransac2_eigensolver_kernel<S, 2>), g2, bl, 0, stream, a);                             \
  }
#else
  (void)split;
#define PNEC_RANSAC_LAUNCH(S) hipLaunchKernelGGL((ransac2_eigensolver_kernel<S, 0>), g1, bl, 0, stream, a);
#endif
  if (scheme == 1) { PNEC_RANSAC_LAUNCH(1) }
  else if (scheme == 2) { PNEC_RANSAC_LAUNCH(2) }
  else { PNEC_RANSAC_LAUNCH(0) }
#undef PNEC_RANSAC_LAUNCH
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) {
    EsBatchArgs b;
    std::memset(&b, 0, sizeof(b));
    b.s = a.scratch;
    b.n_pairs = n_pairs;
    b.data = data;
    b.block_offset = block_offset;
    b.count = count;
    b.out_q = out_q;
    b.out_t = out_t;
    hipStream_t es_stream = stream;
    if (tail_stream) {
      e = hipEventRecord(tail_fork, stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(tail_stream, tail_fork, 0);
      if (e != hipSuccess) return e;
      es_stream = tail_stream;
    }
    launch_es_batch_translation(b, scheme, es_stream);
    e = hipGetLastError();
    if (e == hipSuccess && tail_stream) e = hipEventRecord(tail_done, tail_stream);
  }
#ifdef PNEC_FRONT_DEBUG
  {
    unsigned long long c[24] = {0};
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_dbg), sizeof(c));
    const double per = 1.0 / (4.0 * (double)(n_pairs > 0 ? n_pairs : 1));  // lane counts / 4 = quads, per pair
    std::fprintf(stderr, "ransac minimiser events per pair (quads): warm_evals=%.1f jacobi_fallbacks=%.2f newton_its=%.1f "
                 "levenberg_shifts=%.2f full_step_rejected=%.2f short_step_batches=%.2f wave_evals=%.2f wave_evals_with_fallback=%.2f | as executed by the wavefront: eigen_steps=%.2f levenberg_tries=%.2f heads=%.2f poly_starts=%.2f\n",
                 c[0] * per, c[1] * per, c[2] * per, c[3] * per, c[4] * per, c[5] * per, c[6] * per, c[7] * per,
                 c[8] * per, c[9] * per, c[10] * per, c[11] * per);
    std::fprintf(stderr, "  queue form, per pair: heads as executed=%.2f levenberg tries as executed=%.2f | per quad: heads=%.1f shifts=%.2f\n",
                 c[16] * per, c[17] * per, c[19] * per, c[18] * per);
  }
#endif
  if (a.trace) {
    std::vector<unsigned long long> h(kPhCount * (size_t)n_pairs);
    if (hipStreamSynchronize(stream) == hipSuccess &&
        hipMemcpy(h.data(), a.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      double m[kPhCount] = {0};
      for (int64_t p = 0; p < n_pairs; ++p)
        for (int k = 0; k < kPhCount; ++k) m[k] += (double)h[(size_t)(kPhCount * p + k)];
      if (tr[0] == '/' || tr[0] == '.') {  // a path: the raw [n_pairs, kPhCount] records of this launch (overwritten per launch)
        if (std::FILE *fp = std::fopen(tr, "wb")) {
          std::fwrite(h.data(), sizeof(unsigned long long), h.size(), fp);
          std::fclose(fp);
        }
      }
      static const char *names[kPhCount] = {"sample+sums", "newton", "model", "score", "consume", "inliers", "final_es", "total",
                                            "its_sum16", "its_max_sum (two-pair form: evaluations executed)", "rounds", "evaluations_executed (one-pair form)"};
      std::fprintf(stderr, "ransac_eigensolver phases (mean s_memtime clocks per pair, %lld pairs):", (long long)n_pairs);
      for (int k = 0; k < kPhCount; ++k) std::fprintf(stderr, " %s=%.0f", names[k], m[k] / (double)n_pairs);
      std::fprintf(stderr, "\n");
      if (two && n_pairs > 1) {
        // the launch's timeline from the wavefronts' start / end stamps (two-pair form): how much of it is the tail
        std::vector<unsigned long long> st, en;
        for (int64_t p = 0; p < n_pairs; p += 2) {
          st.push_back(h[(size_t)(kPhCount * p + kRpFinal)]);
          en.push_back(h[(size_t)(kPhCount * p + 11)]);
        }
        const unsigned long long t0 = *std::min_element(st.begin(), st.end());
        const unsigned long long t1 = *std::max_element(en.begin(), en.end());
        std::vector<double> dur(st.size()), end(st.size());
        double busy = 0.0;
        for (size_t i = 0; i < st.size(); ++i) { dur[i] = (double)(en[i] - st[i]); end[i] = (double)(en[i] - t0); busy += dur[i]; }
        std::sort(dur.begin(), dur.end());
        std::sort(end.begin(), end.end());
        std::sort(st.begin(), st.end());
        auto q = [](const std::vector<double> &v, double f) { return v[(size_t)(f * (double)(v.size() - 1))]; };
        std::fprintf(stderr, "  timeline (10 ns ticks, %zu wavefronts): launch %.0f | wavefront p50 %.0f p90 %.0f p99 %.0f max %.0f | done at 50 %% %.0f 90 %% %.0f 99 %% %.0f | last start %.0f | wavefront time / launch = %.0f slots busy on average\n",
                     st.size(), (double)(t1 - t0), q(dur, 0.5), q(dur, 0.9), q(dur, 0.99), dur.back(), q(end, 0.5), q(end, 0.9), q(end, 0.99),
                     (double)(st.back() - t0), busy / (double)(t1 - t0));
      }
    }
    (void)hipFree(a.trace);
  }
  return e;
}

// ---- device self-test of the smallest-eigenpair route (sym_eig3_min_start -> sym_eig3_min_rqi, sweeps as the
// last resort) against the Jacobi sweeps, on positive semi-definite 3x3 matrices of every awkward kind:
// generic, smallest two eigenvalues 1e-6 apart (relative), rank one, rank two, scaled by 1e-12 and 1e+12.
// out[lane]: worst residual |M e - lambda e| / |M|, out[64 + lane]: worst (lambda - lambda_jacobi) / |M|,
// out[128 + lane]: how often the iteration could not vouch for its result (the sweeps decided)
__global__ void eig_selftest_kernel(double *out) {
  const int lane = threadIdx.x;
  unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(lane + 1);
  auto rnd = [&]() {  // uniform in (-1, 1)
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(st >> 11) * (2.0 / 9007199254740992.0) - 1.0;
  };
  double worst_res = 0.0, worst_lam = 0.0, undecided = 0.0;
  for (int trial = 0; trial < 256; ++trial) {
    // an orthonormal basis from two random vectors (Gram-Schmidt), eigenvalues by kind
    double a[3] = {rnd(), rnd(), rnd()}, b[3] = {rnd(), rnd(), rnd()};
    double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) + 1e-300;
    for (int k = 0; k < 3; ++k) a[k] /= na;
    const double ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    for (int k = 0; k < 3; ++k) b[k] -= ab * a[k];
    double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) + 1e-300;
    for (int k = 0; k < 3; ++k) b[k] /= nb;
    const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    const int kind = (trial + lane) % 6;
    double l1 = 1e-5 * (1.0 + rnd() * 0.5), l2 = 1.0 + 0.5 * rnd(), l3 = 2.0 + 0.5 * rnd(), scale = 1.0;
    if (kind == 1) l2 = l1 * (1.0 + 1e-6);       // nearly double smallest eigenvalue
    if (kind == 2) { l1 = 0.0; l2 = 0.0; }       // rank one
    if (kind == 3) l1 = 0.0;                     // rank two
    if (kind == 4) scale = 1e-12;
    if (kind == 5) scale = 1e+12;
    double M[9];
    for (int r = 0; r < 3; ++r)
      for (int q = 0; q < 3; ++q) M[3 * r + q] = scale * (l1 * a[r] * a[q] + l2 * b[r] * b[q] + l3 * c[r] * c[q]);
    double w[3], V[9];
    sym_eig3(M, w, V);
    double e[3] = {0.0, 0.0, 1.0}, lam = 0.0;
    sym_eig3_min_start(M, e);
    bool have = sym_eig3_min_rqi(M, e, lam);
    if (!have) {
      undecided += 1.0;
      lam = w[0];
      e[0] = V[0]; e[1] = V[3]; e[2] = V[6];
    }
    const double norm = fabs(w[2]) + fabs(w[0]) + 1e-300;
    double res = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double me = M[3 * r] * e[0] + M[3 * r + 1] * e[1] + M[3 * r + 2] * e[2] - lam * e[r];
      res = fmax(res, fabs(me));
    }
    worst_res = fmax(worst_res, res / norm);
    worst_lam = fmax(worst_lam, (lam - w[0]) / norm);  // must not be ABOVE the smallest eigenvalue
  }
  out[lane] = worst_res;
  out[64 + lane] = worst_lam;
  out[128 + lane] = undecided;
}

hipError_t launch_frontend_selftest(double *d_out /* 192 doubles */, hipStream_t stream) {
  hipLaunchKernelGGL(eig_selftest_kernel, dim3(1), dim3(kWave), 0, stream, d_out);
  return hipGetLastError();
}

#ifdef PNEC_ISA_PROBE
// -DPNEC_ISA_PROBE (tools/isa_front_regions.py only): the pieces of one evaluation of the eigenvalue function as kernels
// of their own, so that their FP64 instructions can be counted in the assembly: the rotation from the Cayley vector, M from
// the 36 sums, and the whole evaluation with and without its gradient (the difference is the gradient).
__global__ void probe_cayley_kernel(const double *vin, double *out) {
  const double v[3] = {vin[0], vin[1], vin[2]};
  double R[9];
  cayley_to_rot(v, R);
  for (int i = 0; i < 9; ++i) out[i] = R[i];
}
__global__ void probe_m_kernel(const double *G, const double *Rin, double *out) {
  double Gr[36], R[9], M[9];
  for (int i = 0; i < 36; ++i) Gr[i] = G[i];
  for (int i = 0; i < 9; ++i) R[i] = Rin[i];
  sums_to_m(Gr, R, M);
  for (int i = 0; i < 9; ++i) out[i] = M[i];
}
template <bool GRAD>
__global__ void probe_value_grad_kernel(const double *G, const double *vin, double *out) {
  __shared__ double Gs[36];
  if (threadIdx.x < 36) Gs[threadIdx.x] = G[threadIdx.x];
  __syncthreads();
  const double v[3] = {vin[0], vin[1], vin[2]};
  double g[3] = {0.0, 0.0, 0.0}, e[3] = {vin[3], vin[4], vin[5]};
  out[0] = es_value_grad<1>(Gs, v, GRAD ? g : nullptr, nullptr, e, true);
  out[1] = g[0]; out[2] = g[1]; out[3] = g[2]; out[4] = e[0]; out[5] = e[1]; out[6] = e[2];
}
template __global__ void probe_value_grad_kernel<true>(const double *, const double *, double *);
template __global__ void probe_value_grad_kernel<false>(const double *, const double *, double *);
#endif

// ------------------------------------------------------------------------------------------
// host side: launchers called from pnec_capi.hip
static std::mutex g_fib_mutex;
static bool g_fib_ready[64] = {false};

// scf.cc:53-72, including the float division (C6); computed on the host with libm, uploaded once per
// device into the kernel's constant table
hipError_t fibonacci_table(int device) {
  std::lock_guard<std::mutex> lock(g_fib_mutex);
  if (device < 0 || device >= 64) return hipErrorInvalidDevice;
  if (!g_fib_ready[device]) {
    std::vector<double> pts(kFibStride * 500);
    const int samples = 500;
    const double phi = M_PI * (3.0 - std::sqrt(5.0));
    for (int i = 0; i < samples; ++i) {
      const double y = 1.0 - ((float)i / (float)(samples - 1)) * 2.0;
      const double radius = std::sqrt(1 - y * y);
      const double theta = phi * (float)i;
      double *p = &pts[(size_t)kFibStride * i];
      p[0] = std::cos(theta) * radius;
      p[1] = y;
      p[2] = std::sin(theta) * radius;
      // products of the direction, for t'Bt = B00 txx + B11 tyy + B22 tzz + B01 (2 txy) + ...
      p[3] = p[0] * p[0]; p[4] = p[1] * p[1]; p[5] = p[2] * p[2];
      p[6] = 2.0 * p[0] * p[1]; p[7] = 2.0 * p[0] * p[2]; p[8] = 2.0 * p[1] * p[2];
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_fib), pts.data(), sizeof(double) * pts.size());
    if (e != hipSuccess) return e;
    std::vector<float> pts32(pts.begin(), pts.end());
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_fib32), pts32.data(), sizeof(float) * pts32.size());
    if (e != hipSuccess) return e;
    std::vector<double> prod(6 * 512, 0.0);
    for (int i = 0; i < samples; ++i)
      for (int k = 0; k < 6; ++k) prod[(size_t)512 * k + i] = pts[(size_t)kFibStride * i + 3 + k];
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_fib_prod), prod.data(), sizeof(double) * prod.size());
    if (e != hipSuccess) return e;
    g_fib_ready[device] = true;
  }
  return hipSuccess;
}

// the work counters of the current device (PNEC_WORK_COUNT builds; otherwise *compiled_in = 0 and zeros)
hipError_t frontend_work_counters(int reset, unsigned long long *out16, int *compiled_in) {
  for (int i = 0; i < kWkCount; ++i) out16[i] = 0;
#ifdef PNEC_WORK_COUNT
  *compiled_in = 1;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_work), sizeof(unsigned long long) * kWkCount);
  if (e == hipSuccess && reset) {
    const unsigned long long zeros[kWkCount] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_work), zeros, sizeof(zeros));
  }
  return e;
#else
  (void)reset;
  *compiled_in = 0;
  return hipSuccess;
#endif
}

// scratch_d: kFrontScratchDoubles * n_pairs doubles, scratch_i: kFrontScratchInts * n_pairs ints (device)
hipError_t launch_nec_eigensolver(const double *data, const int64_t *block_offset, const int32_t *count,
                                  int64_t n_pairs, const double *init_q, double *out_q, double *out_t,
                                  int32_t *out_iterations, double *scratch_d, int32_t *scratch_i,
                                  hipStream_t stream, int scheme) {
  if (n_pairs <= 0) return hipSuccess;
  FrontArgs a;
  std::memset(&a, 0, sizeof(a));
  a.data = data;
  a.block_offset = block_offset;
  a.count = count;
  a.init_q = init_q;
  const FrontScratch sc = front_scratch(scratch_d, scratch_i, n_pairs);
  hipLaunchKernelGGL(sums36_kernel<false>, dim3((unsigned)n_pairs), dim3(kWave), 0, stream, a, sc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  EsBatchArgs b;
  std::memset(&b, 0, sizeof(b));
  b.s = sc;
  b.n_pairs = n_pairs;
  b.data = data;
  b.block_offset = block_offset;
  b.count = count;
  b.out_q = out_q;
  b.out_t = out_t;
  b.out_iterations = out_iterations;
  launch_es_batch_translation(b, scheme, stream);
  return hipGetLastError();
}

hipError_t launch_weighted_eigensolver(int device, const double *data, const int64_t *block_offset,
                                       const int32_t *count, int64_t n_pairs, int n_max, const double *init_q,
                                       const double *init_t, double reg, int weighted_iterations,
                                       double *out_q, double *out_t, int32_t *out_iterations,
                                       double *scratch_d, int32_t *scratch_i, hipStream_t stream, int scheme) {
  if (n_pairs <= 0) return hipSuccess;
  FrontArgs a;
  std::memset(&a, 0, sizeof(a));
  hipError_t e = fibonacci_table(device);
  if (e != hipSuccess) return e;
  a.data = data;
  a.block_offset = block_offset;
  a.count = count;
  a.init_q = init_q;
  a.init_t = init_t;
  a.out_q = out_q;
  a.out_t = out_t;
  a.out_iterations = out_iterations;
  a.reg = reg;
  a.weighted_iterations = weighted_iterations;
  // the weighted sums of every pair, then the first round's eigenvalue minimisation sixteen pairs per wavefront
  const FrontScratch sc = front_scratch(scratch_d, scratch_i, n_pairs);
  hipLaunchKernelGGL(sums36_kernel<true>, dim3((unsigned)n_pairs), dim3(kWave), 0, stream, a, sc);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  {
    EsBatchArgs b;
    std::memset(&b, 0, sizeof(b));
    b.s = sc;
    b.n_pairs = n_pairs;
#ifdef PNEC_FRONT_DEBUG
    {
      const unsigned long long zeros[4] = {0};
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), zeros, sizeof(zeros), 12 * sizeof(unsigned long long));
    }
#endif
    const int rounds = std::min(weighted_iterations - 1, kEsMaxRounds);  // (the ABI refuses more under scheme 1; 0 and 2 freeze a pair still at its cap)
    if (scheme == 1)
      hipLaunchKernelGGL((es_batch_alt_kernel<kEpiNone, 1>), dim3(es_batch_blocks(n_pairs)), dim3(kWave), 0, stream, b, rounds);
    else if (scheme == 2)
      hipLaunchKernelGGL((es_batch_alt_kernel<kEpiNone, 2>), dim3(es_batch_blocks(n_pairs)), dim3(kWave), 0, stream, b, rounds);
    else
      hipLaunchKernelGGL(es_batch_kernel<kEpiNone>, dim3(es_batch_blocks(n_pairs)), dim3(kWave), 0, stream, b, rounds);
    if ((e = hipGetLastError()) != hipSuccess) return e;
#ifdef PNEC_FRONT_DEBUG
    {
      unsigned long long c[4] = {0};
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_dbg), sizeof(c), 12 * sizeof(unsigned long long));
      unsigned long long who = 0;
      (void)hipMemcpyFromSymbol(&who, HIP_SYMBOL(g_dbg), sizeof(who), 11 * sizeof(unsigned long long));
      std::fprintf(stderr, "weighted stage, first minimisation of %lld pairs: max Newton iterations %llu, at the cap (50): %llu (one of them: pair %llu), >= 20: %llu, mean %.2f\n",
                   (long long)n_pairs, c[0], c[1], who, c[2], (double)c[3] / (double)n_pairs);
    }
#endif
  }
  a.pre_its = sc.its;
  {
    const char *io = std::getenv("PNEC_WES_INDEX_ORDER");  // (A/B: =1 launches in index order)
    a.order = (io && *io && *io != '0') ? nullptr : sc.wo_order;
  }
  a.n_pairs = n_pairs;
  a.pre_v_rounds = sc.v_rounds;
  a.pre_n_es = sc.n_es;
  // PNEC_HIP_TRACE_FRONT=1: per-phase clocks of every pair, averaged and printed to stderr (diagnostics;
  // synchronises, never set it for timed runs)
  const char *tr = std::getenv("PNEC_HIP_TRACE_FRONT");
  if (tr && *tr && n_max <= 64 * kWave) {
    e = hipMalloc(&a.trace, sizeof(unsigned long long) * kPhCount * (size_t)n_pairs);
    if (e != hipSuccess) return e;
  }
  // n_max bounds the pairs' sizes (after an inlier extraction: the source's sizes)
  if (n_max <= 8 * kWave)
    hipLaunchKernelGGL(weighted_eigensolver_kernel<true>, dim3((unsigned)n_pairs), dim3(kWave), 0, stream, a);
  else if (n_max <= 16 * kWave)
    hipLaunchKernelGGL(weighted_eigensolver_mixed_kernel<2>, dim3((unsigned)n_pairs), dim3(2 * kWave), 0, stream, a);
  else
    hipLaunchKernelGGL(weighted_eigensolver_mixed_kernel<8>, dim3((unsigned)n_pairs), dim3(8 * kWave), 0, stream, a);
  e = hipGetLastError();
  if (a.trace) {
    std::vector<unsigned long long> h(kPhCount * (size_t)n_pairs);
    if (hipStreamSynchronize(stream) == hipSuccess &&
        hipMemcpy(h.data(), a.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      double m[kPhCount] = {0};
      for (int64_t p = 0; p < n_pairs; ++p)
        for (int k = 0; k < kPhCount; ++k) m[k] += (double)h[(size_t)(kPhCount * p + k)];
      if (tr[0] == '/' || tr[0] == '.') {  // a path: the raw [n_pairs, kPhCount] records of this launch (overwritten per launch)
        if (std::FILE *fp = std::fopen(tr, "wb")) {
          std::fwrite(h.data(), sizeof(unsigned long long), h.size(), fp);
          std::fclose(fp);
        }
      }
      static const char *names[kPhCount] = {"sums36", "newton", "tables", "search", "cur_cost", "scf", "total", "-", "-", "-", "-", "-"};
      std::fprintf(stderr, "weighted_eigensolver phases (mean s_memtime clocks per pair, %lld pairs):", (long long)n_pairs);
      for (int k = 0; k < kPhTotal + 1; ++k) std::fprintf(stderr, " %s=%.0f", names[k], m[k] / (double)n_pairs);
      std::fprintf(stderr, "\n");
      // where the launch's duration comes from: the distribution over the pairs (a launch ends with its slowest)
      {
        int64_t worst = 0;
        for (int64_t p = 1; p < n_pairs; ++p)
          if (h[(size_t)(kPhCount * p + kPhTotal)] > h[(size_t)(kPhCount * worst + kPhTotal)]) worst = p;
        const unsigned long long *w = &h[(size_t)(kPhCount * worst)];
        std::fprintf(stderr, "  slowest pair %lld: newton %llu tables %llu search %llu scf %llu total %llu | searches %llu candidates %llu "
                     "double-precision evaluations %llu\n", (long long)worst, w[kPhNewton], w[kPhTables], w[kPhSearch], w[kPhScf],
                     w[kPhTotal], w[8], w[9], w[10]);
        std::fprintf(stderr, "  means: searches %.3f candidates %.3f double-precision evaluations %.3f\n", m[8] / (double)n_pairs,
                     m[9] / (double)n_pairs, m[10] / (double)n_pairs);
      }
      for (int k : {(int)kPhSearch, (int)kPhScf, (int)kPhNewton, (int)kPhTotal}) {
        std::vector<unsigned long long> v((size_t)n_pairs);
        for (int64_t p = 0; p < n_pairs; ++p) v[(size_t)p] = h[(size_t)(kPhCount * p + k)];
        std::sort(v.begin(), v.end());
        auto q = [&](double f) { return (double)v[(size_t)std::min<double>((double)n_pairs - 1, f * (double)n_pairs)]; };
        std::fprintf(stderr, "  %-8s p50 %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f\n", names[k], q(0.5), q(0.9), q(0.99),
                     q(0.999), (double)v.back());
      }
    }
    (void)hipFree(a.trace);
  }
  return e;
}

}  // namespace pnec_hip
