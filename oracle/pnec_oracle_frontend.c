/*
 * pnec_oracle_frontend.c -- TEST INFRASTRUCTURE ONLY (same rules as pnec_oracle.c).
 *
 * CPU restatement of the stages in front of the least-squares refinement:
 *   ComposeM / TranslationFromM / Weight        src/common/common.cc:127-136,157-181,183-208
 *   fibonacci_sphere / obj_fun / scf            src/optimization/scf.cc:43-72,109-148
 *   PNEC::Eigensolver (no RANSAC branch)        src/rel_pose_estimation/pnec.cc:231-281
 *   PNEC::WeightedEigensolver                   src/rel_pose_estimation/pnec.cc:283-348
 *   PNEC::Solve                                 src/rel_pose_estimation/pnec.cc:77-124
 *
 * PARITY STATUS
 *   pinned   : fibonacci_sphere, obj_fun (goldens from scripts/pnec/scf.py), Weight and the A_i/B_i
 *              construction (they are the golden-pinned PNEC energy in disguise).
 *   unpinned : opengv::relative_pose::eigensolver.  opengv is not vendored (basalt master,
 *              un-pinned) and the reference has no tests.  The minimiser below
 *              restates the PUBLISHED algorithm (Kneip & Lynen, "Direct optimization of
 *              frame-to-frame rotation", ICCV 2013): minimise the smallest eigenvalue of
 *              M(R) = sum_i (f1_i x R f2_i)(f1_i x R f2_i)' over the Cayley parameters of R,
 *              starting from the supplied rotation.  opengv (as far as it can be recalled without
 *              its source: modules/main.cpp eigensolver_main) walks down the normalised
 *              finite-difference gradient with a step length that doubles while the eigenvalue
 *              falls and halves while it rises, and stops when the step length drops below its
 *              xtol or after 50 iterations -- i.e. it reaches the minimiser only to about its step
 *              tolerance.  Here it is a damped Newton iteration run to a tight tolerance: same
 *              minimiser, different path, smaller stopping slack.  Trajectory parity with opengv
 *              is therefore UNPINNED and cannot be pinned in this image.
 *              Eigenvector signs (TranslationFromM, scf) are arbitrary in Eigen; here: the
 *              component of largest magnitude is made positive.
 *   Reference quirks reproduced: ComposeM skips correspondence 0 (C7); WeightedEigensolver
 *   recomputes its weights from the INITIAL pose every iteration (C3) and scales them by 1e-8
 *   (C4); alt_construct_E drops the -frac*B term (C5); fibonacci_sphere divides in float (C6).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "pnec_oracle.h"

static inline double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void mat_vec(const double R[9], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = R[3 * r] * x[0] + R[3 * r + 1] * x[1] + R[3 * r + 2] * x[2];
}
static inline void matT_vec(const double R[9], const double x[3], double y[3]) {
  for (int r = 0; r < 3; ++r) y[r] = R[r] * x[0] + R[3 + r] * x[1] + R[6 + r] * x[2];
}

/* ---- symmetric 3x3 eigen-decomposition: cyclic Jacobi, eigenvalues ascending ------------- */
/* A row-major symmetric; w ascending; V row-major with eigenvectors in COLUMNS; each column has its
 * largest-magnitude component positive. */
void pnec_oracle_sym_eig3(const double A_in[9], double w[3], double V[9]) {
  double A[9];
  memcpy(A, A_in, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dg = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-34 * dg || off == 0.0) break;
    static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int k = 0; k < 3; ++k) {
      const int p = P[k], q = Q[k];
      const double apq = A[3 * p + q];
      if (apq == 0.0) continue;
      const double theta = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
      /* A <- J' A J */
      for (int r = 0; r < 3; ++r) {
        const double arp = A[3 * r + p], arq = A[3 * r + q];
        A[3 * r + p] = c * arp - s * arq;
        A[3 * r + q] = s * arp + c * arq;
      }
      for (int r = 0; r < 3; ++r) {
        const double apr = A[3 * p + r], aqr = A[3 * q + r];
        A[3 * p + r] = c * apr - s * aqr;
        A[3 * q + r] = s * apr + c * aqr;
      }
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[3 * r + p], vrq = V[3 * r + q];
        V[3 * r + p] = c * vrp - s * vrq;
        V[3 * r + q] = s * vrp + c * vrq;
      }
    }
  }
  double d[3] = {A[0], A[4], A[8]};
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (d[idx[j]] > d[idx[j + 1]]) {
        const int t = idx[j];
        idx[j] = idx[j + 1];
        idx[j + 1] = t;
      }
  double Vs[9];
  for (int c = 0; c < 3; ++c) {
    w[c] = d[idx[c]];
    int big = 0;
    for (int r = 1; r < 3; ++r)
      if (fabs(V[3 * r + idx[c]]) > fabs(V[3 * big + idx[c]])) big = r;
    const double sg = V[3 * big + idx[c]] < 0.0 ? -1.0 : 1.0;
    for (int r = 0; r < 3; ++r) Vs[3 * r + c] = sg * V[3 * r + idx[c]];
  }
  memcpy(V, Vs, sizeof(Vs));
}

/* ---- common.cc:127-136 (loop starts at i = 1 when skip_first) ------------------------------ */
void pnec_oracle_compose_m(int64_t n, const double *bvs1, const double *bvs2, const double R[9],
                           int skip_first, double M[9]) {
  memset(M, 0, 9 * sizeof(double));
  for (int64_t i = skip_first ? 1 : 0; i < n; ++i) {
    double u[3], nn[3];
    mat_vec(R, bvs2 + 3 * i, u);
    cross3(bvs1 + 3 * i, u, nn);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M[3 * r + c] += nn[r] * nn[c];
  }
}

/* ---- common.cc:157-181: unit eigenvector of the smallest eigenvalue ------------------------ */
void pnec_oracle_translation_from_m(const double M[9], double t[3]) {
  double w[3], V[9];
  pnec_oracle_sym_eig3(M, w, V);
  const double v[3] = {V[0], V[3], V[6]};
  const double nv = sqrt(dot3(v, v));
  for (int k = 0; k < 3; ++k) t[k] = v[k] / nv;
}

/* ---- common.cc:183-208; cov column-major 9 --------------------------------------------------- */
double pnec_oracle_weight(const double f1[3], const double f2[3], const double t[3], const double R[9],
                          const double *cov, double reg, int host_frame) {
  double v[3];
  if (host_frame) {
    double p[3];
    mat_vec(R, f2, p);
    cross3(t, p, v); /* (t' [R f2]x)' = -(R f2) x t ... sign irrelevant in the quadratic form */
    double q = 0.0;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) q += v[r] * cov[3 * c + r] * v[c];
    return 1.0 / q;
  }
  double m[3];
  cross3(t, f1, m);
  matT_vec(R, m, v);
  double q = 0.0;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) q += v[r] * cov[3 * c + r] * v[c];
  return 1.0 / (q + reg);
}

/* ---- Cayley parameterisation (opengv math/cayley, published formulas) ----------------------- */
void pnec_oracle_cayley_to_rot(const double v[3], double R[9]) {
  const double x = v[0], y = v[1], z = v[2];
  const double s = 1.0 / (1.0 + x * x + y * y + z * z);
  R[0] = s * (1 + x * x - y * y - z * z); R[1] = s * 2 * (x * y - z); R[2] = s * 2 * (x * z + y);
  R[3] = s * 2 * (x * y + z); R[4] = s * (1 - x * x + y * y - z * z); R[5] = s * 2 * (y * z - x);
  R[6] = s * 2 * (x * z - y); R[7] = s * 2 * (y * z + x); R[8] = s * (1 - x * x - y * y + z * z);
}
void pnec_oracle_rot_to_cayley(const double R[9], double v[3]) {
  /* C = (R - I)(R + I)^-1 is skew; v = (-C(1,2), C(0,2), -C(0,1)) */
  double A[9], B[9];
  for (int i = 0; i < 9; ++i) {
    A[i] = R[i] - (i % 4 == 0 ? 1.0 : 0.0);
    B[i] = R[i] + (i % 4 == 0 ? 1.0 : 0.0);
  }
  /* inverse of B by adjugate */
  const double c00 = B[4] * B[8] - B[5] * B[7], c01 = B[5] * B[6] - B[3] * B[8], c02 = B[3] * B[7] - B[4] * B[6];
  const double det = B[0] * c00 + B[1] * c01 + B[2] * c02;
  double Bi[9];
  Bi[0] = c00 / det; Bi[1] = (B[2] * B[7] - B[1] * B[8]) / det; Bi[2] = (B[1] * B[5] - B[2] * B[4]) / det;
  Bi[3] = c01 / det; Bi[4] = (B[0] * B[8] - B[2] * B[6]) / det; Bi[5] = (B[2] * B[3] - B[0] * B[5]) / det;
  Bi[6] = c02 / det; Bi[7] = (B[1] * B[6] - B[0] * B[7]) / det; Bi[8] = (B[0] * B[4] - B[1] * B[3]) / det;
  double Cm[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Cm[3 * r + c] = A[3 * r] * Bi[c] + A[3 * r + 1] * Bi[3 + c] + A[3 * r + 2] * Bi[6 + c];
  v[0] = -Cm[5];
  v[1] = Cm[2];
  v[2] = -Cm[1];
}

/* ---- eigenvalue minimisation ----------------------------------------------------------------- */
typedef struct {
  int64_t n;
  const double *b1, *b2;
} es_data;

/* lambda_min(M(R(v))), its eigenvector e, and d lambda / d v (analytic: e' dM e) */
static double g_es_trace; /* trace of the last composed M (noise floor of the line search) */
#pragma omp threadprivate(g_es_trace)
static double es_value_grad(const es_data *D, const double v[3], double g[3]) {
  double R[9], M[9], w[3], V[9];
  pnec_oracle_cayley_to_rot(v, R);
  pnec_oracle_compose_m(D->n, D->b1, D->b2, R, 0, M);
  g_es_trace = M[0] + M[4] + M[8];
  pnec_oracle_sym_eig3(M, w, V);
  if (!g) return w[0];
  const double e[3] = {V[0], V[3], V[6]};
  /* dR/dv_k = (dN/dv_k - 2 v_k R) / s,  N = (1-|v|^2) I + 2[v]x + 2 v v' */
  const double s = 1.0 + dot3(v, v);
  for (int k = 0; k < 3; ++k) {
    double dN[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) dN[4 * i] = -2.0 * v[k];
    /* 2 [e_k]x */
    const int a = (k + 1) % 3, b = (k + 2) % 3;
    dN[3 * b + a] += 2.0;
    dN[3 * a + b] -= 2.0;
    for (int i = 0; i < 3; ++i) {
      dN[3 * k + i] += 2.0 * v[i];
      dN[3 * i + k] += 2.0 * v[i];
    }
    double dR[9];
    for (int i = 0; i < 9; ++i) dR[i] = (dN[i] - 2.0 * v[k] * R[i]) / s;
    double acc = 0.0;
    for (int64_t i = 0; i < D->n; ++i) {
      double u[3], nn[3], du[3], dn[3];
      mat_vec(R, D->b2 + 3 * i, u);
      cross3(D->b1 + 3 * i, u, nn);
      mat_vec(dR, D->b2 + 3 * i, du);
      cross3(D->b1 + 3 * i, du, dn);
      acc += 2.0 * dot3(e, nn) * dot3(e, dn);
    }
    g[k] = acc;
  }
  return w[0];
}

static int solve3_spd(const double H[9], const double b[3], double x[3]) {
  /* Cholesky of a symmetric 3x3; 0 if not positive definite */
  const double l00s = H[0];
  if (!(l00s > 0.0)) return 0;
  const double l00 = sqrt(l00s), l10 = H[3] / l00, l20 = H[6] / l00;
  const double l11s = H[4] - l10 * l10;
  if (!(l11s > 0.0)) return 0;
  const double l11 = sqrt(l11s), l21 = (H[7] - l20 * l10) / l11;
  const double l22s = H[8] - l20 * l20 - l21 * l21;
  if (!(l22s > 0.0)) return 0;
  const double l22 = sqrt(l22s);
  const double z0 = b[0] / l00, z1 = (b[1] - l10 * z0) / l11, z2 = (b[2] - l20 * z0 - l21 * z1) / l22;
  x[2] = z2 / l22;
  x[1] = (z1 - l21 * x[2]) / l11;
  x[0] = (z0 - l10 * x[1] - l20 * x[2]) / l00;
  return 1;
}

/* ---- which iteration the eigenvalue minimisations run ------------------------------------------------------------------
 * 0 (default): the damped Newton iteration below on lambda_min(M(R(v))) -- the device's default, converged to ~1e-12 rad.
 * 1, 2 [EXT, from memory, unpinned]: the two recollections of opengv's own iteration restated in pnec_oracle_opengv.c
 *   (1: normalised steepest descent with an adaptive step, stops ~1e-5 rad short; 2: Eigen's Levenberg-Marquardt on the
 *   gradient of lambda_min composed with the REDUCED Cayley rotation -- what eigensolver_main is recalled to run).
 * The scheme applies wherever scheme 0 runs the Newton iteration (RANSAC hypotheses, the eigensolver on the inliers,
 * the plain eigensolver, the weighted stage's rounds); schemes 1 and 2 work on the 36 sums, as opengv does.  Test
 * tooling; a process-wide switch: set it before, not during, a batch call. */
static int g_es_scheme = 0;
void pnec_oracle_set_eigensolver_scheme(int scheme) { g_es_scheme = scheme; }
/* RANSAC under scheme 0 with the rules of round 3 ("frozen"): every hypothesis is scored, whatever its minimisation did
 * (opengv scores every model), and a hypothesis' minimisation may take the 50 iterations of any other.  OFF by default:
 * the device's rule since round 4 -- a hypothesis still moving after 25 iterations yields no model -- is what the parity
 * tests compare with; ON to measure what that rule changes (tools/verify_eigensolver_schemes.py). */
static int g_ransac_frozen_rules = 0;
void pnec_oracle_set_ransac_frozen_rules(int on) { g_ransac_frozen_rules = on; }
int pnec_oracle_get_eigensolver_scheme(void) { return g_es_scheme; }
/* RANSAC's chained starts [EXT, from memory, unpinned]: opengv's EigensolverSacProblem::getSelectedDistancesToModel writes
 * the model it scores into the adapter (_adapter.sett12(model.translation); _adapter.setR12(model.rotation)) before it
 * triangulates, and computeModelCoefficients starts every hypothesis from _adapter.getR12() + jitter -- so hypothesis
 * h + 1 starts from the rotation of the last model SCORED (hypothesis h's, whether or not it became the best), not from
 * the initial rotation (the call site: pnec.cc:235-252 hands the adapter the initial rotation once).  OFF by default: the
 * device evaluates a round's sixteen hypotheses side by side from the initial rotation; ON restates the sequential
 * dependence (the device: PNEC_HIP_RANSAC_CHAINED_STARTS, one hypothesis per round).  Statistical either way under
 * opengv's rand(); the difference is which basin a contaminated sample's minimisation starts near. */
static int g_ransac_chained_starts = 0;
void pnec_oracle_set_ransac_chained_starts(int on) { g_ransac_chained_starts = on; }
static int g_es_info; /* scheme 2: Eigen's status of the calling thread's last minimisation (5 = maxfev reached) */
static int g_es_nfev;
#pragma omp threadprivate(g_es_info, g_es_nfev)
int pnec_oracle_es_last_info(void) { return g_es_info; }
int pnec_oracle_es_last_nfev(void) { return g_es_nfev; }

/* The Levenberg shift of an iteration whose Hessian is not positive definite: mu = 2 |x|, x = a lower bound of the
 * Hessian's smallest eigenvalue that is within a few percent of it unless eigenvalues nearly coincide -- three Newton
 * steps on the characteristic polynomial from the Gershgorin bound (from the left of the smallest root the iteration
 * rises monotonically and never passes it).  H + mu I then has its smallest eigenvalue at about |lambda_min|: neither
 * nearly singular nor over-damped.  (Until round 4 the shift was searched in decades -- 0, 1e-6 tr, 1e-5 tr, ... until
 * the factorisation went through -- which lands anywhere between one and ten times |lambda_min|; a minimisation that
 * starts on the flank of a saddle then crawls with steps of g / mu for twenty or thirty iterations.  Counted here on
 * 2 400 RANSAC hypotheses of the benchmark's data, in evaluations as the device spends them: 9.96 -> 9.18 per
 * minimisation, 22 and more 2.6 % -> 0.7 %; the rare long one sets the length of a round of 32 on the device.)  Should
 * the factorisation still fail (rounding, a Hessian with a NaN), the shift grows in decades as before.  Same rule on the
 * device (levenberg_direction). */
#define ES_LEVENBERG_GROWTH 10.0
/* Most Newton iterations of one minimisation: 50, and 25 for the minimisation of a RANSAC hypothesis (the device's
 * kNewtonMaxIterations / kHypothesisMaxIterations).  The hypotheses that get that far are contaminated samples crawling
 * along the flank of a saddle or running off to the minimum at infinity of the Cayley chart; on the device one of them sets
 * the length of its round.  A hypothesis cut off there yields no model (pnec_oracle_ransac_eigensolver). */
#define ES_MAX_ITERATIONS 50
#define ES_HYPOTHESIS_MAX_ITERATIONS 25
/* A full undamped Newton step shorter than this ends a minimisation (quadratic convergence: the point it leads to is within
 * ~C * step^2 of the minimiser).  The minimisation of a RANSAC hypothesis has its own constant (the device's
 * kHypothesisStepDone, where the measurement with 1e-4 is written down); both are 1e-6. */
#define ES_STEP_DONE 1e-6
#define ES_HYPOTHESIS_STEP_DONE 1e-6
static double es_hessian_floor(const double H[9]) {
  const double m00 = H[0], m01 = H[1], m02 = H[2], m11 = H[4], m12 = H[5], m22 = H[8];
  const double trh = m00 + m11 + m22;
  const double c2 = (m00 * m11 - m01 * m01) + (m00 * m22 - m02 * m02) + (m11 * m22 - m12 * m12);
  const double det = m00 * (m11 * m22 - m12 * m12) - m01 * (m01 * m22 - m12 * m02) + m02 * (m01 * m12 - m11 * m02);
  double x = fmin(m00 - fabs(m01) - fabs(m02), fmin(m11 - fabs(m01) - fabs(m12), m22 - fabs(m02) - fabs(m12)));
  for (int q = 0; q < 3; ++q) {
    const double pq = ((x - trh) * x + c2) * x - det, dq = (3.0 * x - 2.0 * trh) * x + c2;
    if (!(dq > 0.0)) break;
    x -= pq / dq;
  }
  return x;
}

/* Damped Newton on v (Cayley): Hessian by forward differences of the analytic gradient
 * (h = 1e-6), Levenberg shift (above) when it is not positive definite and descending, Armijo backtracking.
 * Stops when |step|_inf < 1e-12 (step_done for a full undamped Newton step), |grad|_inf < 1e-14 * (1 + |lambda|) * n, or after
 * max_it iterations. */
static int g_es_trips; /* evaluations of the last minimisation as the device's quad spends them (diagnostics, below) */
#pragma omp threadprivate(g_es_trips)
/* One trip of the device's quad = the point with its three Hessian probes, or four step lengths of the Armijo search:
 * 1 for the start, 1 per iteration whose full step is taken, and ceil(j / 4) + 1 more for one that is cut back j times.
 * Test tooling for tools/sim_ransac_queue.py (how a round's minimisations pack onto sixteen quads). */
int pnec_oracle_es_last_trips(void) { return g_es_trips; }
static int eigensolver_cayley_tol(const es_data *Dp, double v[3], double step_done, int max_it) {
  g_es_trips = 0;
  g_es_info = 0;
  if (g_es_scheme == 1 || g_es_scheme == 2) {
    double G[36];
    pnec_oracle_sums36(Dp->n, Dp->b1, Dp->b2, G);
    if (g_es_scheme == 1) {
      if (pnec_oracle_get_eigensolver_restart()) return pnec_oracle_es_descent_restarts(G, v, 1, 0, NULL);
      return pnec_oracle_es_descent(G, v, &g_es_trips);
    }
    return pnec_oracle_es_lm(G, v, &g_es_nfev, &g_es_info);
  }
  const es_data D = *Dp;
  const int64_t n = D.n;
  double g[3];
  double f = es_value_grad(&D, v, g);
  int it = 0;
  g_es_trips = 1;
  for (; it < max_it; ++it) {
    const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
    if (gmax <= 1e-14 * (1.0 + fabs(f)) * (double)(n > 0 ? n : 1)) break;
    double H[9];
    const double h = 1e-6;
    for (int k = 0; k < 3; ++k) {
      double vp[3] = {v[0], v[1], v[2]}, gp[3];
      vp[k] += h;
      es_value_grad(&D, vp, gp);
      for (int r = 0; r < 3; ++r) H[3 * r + k] = (gp[r] - g[r]) / h;
    }
    for (int r = 0; r < 3; ++r)
      for (int c = r + 1; c < 3; ++c) H[3 * r + c] = H[3 * c + r] = 0.5 * (H[3 * r + c] + H[3 * c + r]);
    double mu = 0.0, d[3];
    const double tr = fabs(H[0]) + fabs(H[4]) + fabs(H[8]);
    int ok = 0;
    for (int tries = 0; tries < 40; ++tries) {
      double Hm[9];
      memcpy(Hm, H, sizeof(Hm));
      Hm[0] += mu; Hm[4] += mu; Hm[8] += mu;
      const double mg[3] = {-g[0], -g[1], -g[2]};
      if (solve3_spd(Hm, mg, d) && dot3(d, g) < 0.0) { ok = 1; break; }
      mu = (tries == 0) ? fmax(2.0 * fmax(-es_hessian_floor(H), 0.0), 1e-6 * (tr + 1e-300)) : mu * ES_LEVENBERG_GROWTH;
    }
    if (!ok) break;
    double alpha = 1.0, fn = f, vn[3];
    const double slope = dot3(d, g);
    int moved = 0, ls = 0;
    for (; ls < 40; ++ls) {
      for (int k = 0; k < 3; ++k) vn[k] = v[k] + alpha * d[k];
      fn = es_value_grad(&D, vn, NULL);
      /* Armijo with a rounding-noise floor: lambda_min carries ~eps * trace(M) of error */
      if (fn <= f + 1e-4 * alpha * slope + 4e-16 * g_es_trace) { moved = 1; break; }
      alpha *= 0.5;
    }
    g_es_trips += 1 + (ls > 0 ? (ls + 3) / 4 + (moved ? 1 : 0) : 0);
    if (!moved) break;
    const double smax = alpha * fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
    memcpy(v, vn, sizeof(vn));
    f = es_value_grad(&D, v, g);
    if (smax < 1e-12) { ++it; break; }
    /* a full undamped Newton step this short: quadratic convergence has put the new point within ~1e-12 of the minimiser
     * (the device stops here without the confirming evaluation; kNewtonStepDone) */
    if (mu == 0.0 && alpha == 1.0 && smax < step_done) { ++it; break; }
  }
  return it;
}
static int eigensolver_cayley(const es_data *Dp, double v[3]) { return eigensolver_cayley_tol(Dp, v, ES_STEP_DONE, ES_MAX_ITERATIONS); }

int pnec_oracle_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                            double R_out[9], int32_t *iterations) {
  es_data D = {n, bvs1, bvs2};
  double v[3];
  pnec_oracle_rot_to_cayley(R0, v);
  const int it = eigensolver_cayley(&D, v);
  pnec_oracle_cayley_to_rot(v, R_out);
  if (iterations) *iterations = it;
  return 0;
}

/* ---- scf.cc --------------------------------------------------------------------------------- */
void pnec_oracle_fibonacci_sphere(int samples, double *pts /* [samples,3] */) {
  const double phi = M_PI * (3.0 - sqrt(5.0));
  for (int i = 0; i < samples; ++i) {
    const double y = 1.0 - ((float)i / (float)(samples - 1)) * 2.0; /* float division (scf.cc:59) */
    const double radius = sqrt(1 - y * y);
    const double theta = phi * (float)i;
    pts[3 * i] = cos(theta) * radius;
    pts[3 * i + 1] = y;
    pts[3 * i + 2] = sin(theta) * radius;
  }
}

static double quad9(const double *A /*row-major sym*/, const double t[3]) {
  double q = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) q += t[r] * A[3 * r + c] * t[c];
  return q;
}

/* scf.cc:43-51; Ai, Bi arrays of n row-major 3x3 */
double pnec_oracle_obj_fun(const double t[3], int64_t n, const double *Ai, const double *Bi) {
  double cost = 0.0;
  for (int64_t i = 0; i < n; ++i) cost += quad9(Ai + 9 * i, t) / quad9(Bi + 9 * i, t);
  return cost;
}

/* scf.cc:128-148 with alt_construct_E's resize/push_back slip (C5): E = sum A_i / (t'B_i t) */
static void scf_steps(int64_t n, const double *Ai, const double *Bi, const double t0[3], int steps,
                      int stop_at_fixed_point, double t_out[3]);
void pnec_oracle_scf(int64_t n, const double *Ai, const double *Bi, const double t0[3], int steps,
                     double t_out[3]) {
  scf_steps(n, Ai, Bi, t0, steps, 0, t_out);
}

/* A_i = n n', B_i = f1hat R Sigma R' f1hat' + reg I   (pnec.cc:317-328); row-major outputs */
void pnec_oracle_build_ab(int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                          const double R[9], double reg, double *Ai, double *Bi) {
  for (int64_t i = 0; i < n; ++i) {
    const double *f1 = bvs1 + 3 * i, *C = covs + 9 * i;
    double u[3], nn[3];
    mat_vec(R, bvs2 + 3 * i, u);
    cross3(f1, u, nn);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Ai[9 * i + 3 * r + c] = nn[r] * nn[c];
    /* P = f1hat R (row-major): P(r,c) = sum_k F(r,k) R(k,c) */
    const double F[9] = {0, -f1[2], f1[1], f1[2], 0, -f1[0], -f1[1], f1[0], 0};
    double P[9], PS[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) P[3 * r + c] = F[3 * r] * R[c] + F[3 * r + 1] * R[3 + c] + F[3 * r + 2] * R[6 + c];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) /* (P Sigma)(r,c), Sigma column-major: S(k,c) = C[3c+k] */
        PS[3 * r + c] = P[3 * r] * C[3 * c] + P[3 * r + 1] * C[3 * c + 1] + P[3 * r + 2] * C[3 * c + 2];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        Bi[9 * i + 3 * r + c] = PS[3 * r] * P[3 * c] + PS[3 * r + 1] * P[3 * c + 1] + PS[3 * r + 2] * P[3 * c + 2] +
                                (r == c ? reg : 0.0);
  }
}

/* ---- pnec.cc:231-281, use_ransac_ = false branch -------------------------------------------- */
void pnec_oracle_nec_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                                 double R_out[9], double t_out[3]) {
  pnec_oracle_eigensolver(n, bvs1, bvs2, R0, R_out, NULL);
  double M[9];
  pnec_oracle_compose_m(n, bvs1, bvs2, R_out, 1, M);
  pnec_oracle_translation_from_m(M, t_out);
}

/* ---- pnec.cc:283-348 --------------------------------------------------------------------------
 * LITERAL restatement of PNEC::WeightedEigensolver: for each of the weighted_iterations - 1 rounds
 *   weights from the INITIAL pose (pnec.cc:297-301, quirk C3) x 1e-8 (C4), bvs2 scaled by sqrt(w)
 *   (:304-308), adapter started at rel_pose.rotationMatrix() (:310-311), eigensolver (:315), A_i, B_i
 *   from the new rotation (:317-328), the current translation against 500 Fibonacci directions
 *   (:330-340), scf with exactly 10 steps (:342-343, scf.cc:128-148), rel_pose <- (R, t) (:345).
 * Every round calls the eigensolver and every scf call runs all of its steps, as the reference reads.
 *
 * device_early_exits != 0 is NOT the reference: it reproduces the two early exits the DEVICE kernel
 * takes (declared deviations) so that tests can separate "the kernel's arithmetic differs
 * from this file's" (tight tolerance, against this twin) from "the early exits change the result"
 * (bounded, measured against the literal form: profiles/r02_frontend_literal_parity.json):
 *   (1) once an eigensolver call has ended below its iteration cap the rotation is kept for the
 *       remaining rounds (the weights never change, C3, so later calls restart at the optimum);
 *   (2) scf stops when its iterate repeats to 4e-15 instead of running all 10 steps. */
static void scf_steps(int64_t n, const double *Ai, const double *Bi, const double t0[3], int steps,
                      int stop_at_fixed_point, double t_out[3]) {
  double t[3] = {t0[0], t0[1], t0[2]};
  for (int s = 0; s < steps; ++s) {
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
      const double w = 1.0 / quad9(Bi + 9 * i, t);
      for (int k = 0; k < 9; ++k) E[k] += w * Ai[9 * i + k];
    }
    double w3[3], V[9];
    pnec_oracle_sym_eig3(E, w3, V);
    const double moved = fmax(fabs(V[0] - t[0]), fmax(fabs(V[3] - t[1]), fabs(V[6] - t[2])));
    t[0] = V[0]; t[1] = V[3]; t[2] = V[6];
    if (stop_at_fixed_point && moved <= 4e-15) break;
  }
  memcpy(t_out, t, 3 * sizeof(double));
}

void pnec_oracle_weighted_eigensolver_ex(int64_t n, const double *bvs1, const double *bvs2,
                                         const double *covs, const double R_init[9], const double t_init[3],
                                         double reg, int weighted_iterations, int device_early_exits,
                                         double R_out[9], double t_out[3]) {
  double R[9], t[3], v[3];
  memcpy(R, R_init, sizeof(R));
  memcpy(t, t_init, sizeof(t));
  pnec_oracle_rot_to_cayley(R_init, v);
  double *w2 = (double *)malloc(sizeof(double) * 3 * (size_t)(n > 0 ? n : 1));
  double *Ai = (double *)malloc(sizeof(double) * 9 * (size_t)(n > 0 ? n : 1));
  double *Bi = (double *)malloc(sizeof(double) * 9 * (size_t)(n > 0 ? n : 1));
  double fib[1500];
  pnec_oracle_fibonacci_sphere(500, fib);
  int rotation_final = 0;
  for (int it = 0; it + 1 < weighted_iterations; ++it) {
    /* weights from the INITIAL pose every iteration (C3), scaled by 1e-8 (C4) */
    for (int64_t i = 0; i < n; ++i) {
      const double w = pnec_oracle_weight(bvs1 + 3 * i, bvs2 + 3 * i, t_init, R_init, covs + 9 * i, reg, 0) * 1e-8;
      const double sw = sqrt(w);
      for (int k = 0; k < 3; ++k) w2[3 * i + k] = bvs2[3 * i + k] * sw;
    }
    double Rn[9];
    {
      es_data D = {n, bvs1, w2};
      if (device_early_exits && g_es_scheme != 1) {
        /* the Cayley vector is carried between rounds, early exit (1): a call that ended for any reason other than its
         * cap (scheme 0: 50 Newton iterations; scheme 2: MINPACK's maxfev) has found the optimum of a function that
         * never changes (C3).  Scheme 1 has no such exit on the device either: its descent stops ~1e-5 short, and every
         * further call moves the rotation a little closer, as in the reference's literal loop */
        if (!rotation_final) {
          const int its = eigensolver_cayley(&D, v);
          rotation_final = g_es_scheme == 2 ? g_es_info != 5 : its < ES_MAX_ITERATIONS;
        }
      } else {
        /* pnec.cc:310-315: a new adapter holding rel_pose's rotation MATRIX, a new eigensolver call */
        pnec_oracle_rot_to_cayley(R, v);
        eigensolver_cayley(&D, v);
      }
      pnec_oracle_cayley_to_rot(v, Rn);
    }
    pnec_oracle_build_ab(n, bvs1, bvs2, covs, Rn, reg, Ai, Bi);
    double best[3] = {t[0], t[1], t[2]};
    double best_cost = pnec_oracle_obj_fun(best, n, Ai, Bi);
    for (int k = 0; k < 500; ++k) {
      const double c = pnec_oracle_obj_fun(fib + 3 * k, n, Ai, Bi);
      if (c < best_cost) {
        best_cost = c;
        memcpy(best, fib + 3 * k, sizeof(best));
      }
    }
    scf_steps(n, Ai, Bi, best, 10, device_early_exits, t);
    memcpy(R, Rn, sizeof(R));
  }
  memcpy(R_out, R, sizeof(R));
  memcpy(t_out, t, sizeof(t));
  free(w2);
  free(Ai);
  free(Bi);
}

/* OpenMP over pairs (test tooling: thousands of pairs against the device) */
void pnec_oracle_weighted_eigensolver_batch(int64_t n_pairs, const int64_t *offsets, const double *bvs1,
                                            const double *bvs2, const double *covs, const double *R_init,
                                            const double *t_init, double reg, int weighted_iterations,
                                            int device_early_exits, int num_threads, double *R_out,
                                            double *t_out) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t a = offsets[p], n = offsets[p + 1] - offsets[p];
    pnec_oracle_weighted_eigensolver_ex(n, bvs1 + 3 * a, bvs2 + 3 * a, covs + 9 * a, R_init + 9 * p, t_init + 3 * p,
                                        reg, weighted_iterations, device_early_exits, R_out + 9 * p, t_out + 3 * p);
  }
}

/* the reference's behaviour: the literal form */
void pnec_oracle_weighted_eigensolver(int64_t n, const double *bvs1, const double *bvs2,
                                      const double *covs, const double R_init[9], const double t_init[3],
                                      double reg, int weighted_iterations, double R_out[9],
                                      double t_out[3]) {
  pnec_oracle_weighted_eigensolver_ex(n, bvs1, bvs2, covs, R_init, t_init, reg, weighted_iterations, 0, R_out,
                                      t_out);
}

/* ---- RANSAC around the eigensolver: pnec.cc:239-272 -------------------------------------------
 * opengv::sac::Ransac<EigensolverSacProblem> restated from its published behaviour (opengv is not
 * in the tree): hypotheses from `sample_size` random correspondences with the start rotation
 * jittered by +-0.01 in Cayley space, eigensolver on the sample, translation = eigenvector of the
 * smallest eigenvalue signed by the directional evidence sum t.(f1 - R f2), score per
 * correspondence = (1 - f1.reproj1) + (1 - f2.reproj2) of the midpoint triangulation, inlier if
 * score < threshold (1e-6 in the reference), adaptive iteration bound k = log(1-p)/log(1-w^s) with
 * p = 0.99, then the eigensolver re-run on the inliers (optimizeModelCoefficients) and the
 * reference's own TranslationFromM(ComposeM(inliers)).  opengv draws from rand(); here every draw
 * is a counter-based hash of (seed, pair, hypothesis, draw), so CPU oracle and device agree. */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
double pnec_oracle_rng_uniform(uint64_t seed, uint64_t pair, uint64_t hyp, uint64_t draw) {
  const uint64_t h = splitmix64(splitmix64(splitmix64(seed ^ 0xD1B54A32D192ED03ull) + pair) + (hyp << 20) + draw);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0); /* [0,1) */
}

/* reprojection score of one correspondence under (R, t) -- opengv midpoint triangulation */
double pnec_oracle_reprojection_score(const double f1[3], const double f2[3], const double R[9],
                                      const double t[3]) {
  double f2u[3];
  mat_vec(R, f2, f2u);
  const double b0 = dot3(t, f1), b1 = dot3(t, f2u);
  const double a00 = dot3(f1, f1), a10 = dot3(f1, f2u), a01 = -a10, a11 = -dot3(f2u, f2u);
  const double det = a00 * a11 - a01 * a10;
  const double l0 = (a11 * b0 - a01 * b1) / det, l1 = (-a10 * b0 + a00 * b1) / det;
  double p[3], p2[3], d[3];
  for (int k = 0; k < 3; ++k) p[k] = 0.5 * (l0 * f1[k] + t[k] + l1 * f2u[k]);
  for (int k = 0; k < 3; ++k) d[k] = p[k] - t[k];
  matT_vec(R, d, p2); /* R' (p - t) */
  const double n1 = sqrt(dot3(p, p)), n2 = sqrt(dot3(p2, p2));
  return (1.0 - dot3(f1, p) / n1) + (1.0 - dot3(f2, p2) / n2);
}

static void es_model_translation(int64_t m, const double *b1, const double *b2, const double R[9], double t[3]) {
  double M[9], w[3], V[9];
  pnec_oracle_compose_m(m, b1, b2, R, 0, M);
  pnec_oracle_sym_eig3(M, w, V);
  t[0] = V[0]; t[1] = V[3]; t[2] = V[6];
  double ev = 0.0; /* directional evidence */
  for (int64_t i = 0; i < m; ++i) {
    double u[3];
    mat_vec(R, b2 + 3 * i, u);
    for (int k = 0; k < 3; ++k) ev += t[k] * (b1[3 * i + k] - u[k]);
  }
  if (ev < 0.0)
    for (int k = 0; k < 3; ++k) t[k] = -t[k];
}

int pnec_oracle_ransac_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                                   uint64_t seed, uint64_t pair_id, int max_iterations, int sample_size,
                                   double threshold, double R_out[9], double t_out[3], uint8_t *inlier_mask,
                                   int32_t *n_inliers, int32_t *iterations) {
  if (sample_size > 16) sample_size = 16;
  if (n < sample_size || sample_size < 1) { /* cannot sample: plain eigensolver, everything an inlier */
    pnec_oracle_nec_eigensolver(n, bvs1, bvs2, R0, R_out, t_out);
    for (int64_t i = 0; i < n; ++i) inlier_mask[i] = 1;
    *n_inliers = (int32_t)n;
    if (iterations) *iterations = 0;
    return 0;
  }
  double v0[3];
  pnec_oracle_rot_to_cayley(R0, v0);
  double best_R[9], best_t[3] = {0, 0, 1};
  memcpy(best_R, R0, sizeof(best_R));
  int best_count = -1, it = 0;
  double k = 1.0;
  while (it < k) {
    int sel[16], m = 0;
    uint64_t draw = 0;
    while (m < sample_size) {
      int64_t idx = (int64_t)(pnec_oracle_rng_uniform(seed, pair_id, (uint64_t)it, draw++) * (double)n);
      if (idx >= n) idx = n - 1;
      int dup = 0;
      for (int j = 0; j < m; ++j) dup |= (sel[j] == idx);
      if (!dup) sel[m++] = (int)idx;
    }
    double s1[48], s2[48], v[3], R[9], t[3];
    for (int j = 0; j < sample_size; ++j) {
      memcpy(s1 + 3 * j, bvs1 + 3 * sel[j], 3 * sizeof(double));
      memcpy(s2 + 3 * j, bvs2 + 3 * sel[j], 3 * sizeof(double));
    }
    for (int c = 0; c < 3; ++c)
      v[c] = v0[c] + (pnec_oracle_rng_uniform(seed, pair_id, (uint64_t)it, 1000 + c) - 0.5) * 2.0 * 0.01;
    es_data D = {sample_size, s1, s2};
    const int hyp_cap = g_ransac_frozen_rules ? ES_MAX_ITERATIONS : ES_HYPOTHESIS_MAX_ITERATIONS;
    const int newton_its = eigensolver_cayley_tol(&D, v, ES_HYPOTHESIS_STEP_DONE, hyp_cap);
    pnec_oracle_cayley_to_rot(v, R);
    es_model_translation(sample_size, s1, s2, R, t);
    /* a minimisation that was cut off yields no model: the rule consumes the hypothesis with a count of zero (why: the
     * device's kHypothesisMaxIterations -- cut off, two floating-point realisations of the iteration stand at different
     * points of a walk that did not converge, and would score differently) */
    int count = 0;
    const int cut_off = !g_ransac_frozen_rules && g_es_scheme == 0 && newton_its >= hyp_cap; /* (schemes 1, 2: every hypothesis
                                                                                              is scored, as opengv does) */
    for (int64_t i = 0; i < n && !cut_off; ++i)
      count += pnec_oracle_reprojection_score(bvs1 + 3 * i, bvs2 + 3 * i, R, t) < threshold;
    if (g_ransac_chained_starts && !cut_off) pnec_oracle_rot_to_cayley(R, v0); /* the scored model stays in the adapter */
    if (count > best_count) {
      best_count = count;
      memcpy(best_R, R, sizeof(R));
      memcpy(best_t, t, sizeof(t));
      const double w = (double)count / (double)n;
      double p_no = 1.0 - pow(w, (double)sample_size);
      p_no = fmax(2.220446049250313e-16, p_no);
      p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
      k = log(1.0 - 0.99) / log(p_no);
    }
    ++it;
    if (it > max_iterations) break;
  }
  /* inliers of the best model, then optimizeModelCoefficients on them */
  int32_t cnt = 0;
  double *i1 = (double *)malloc(sizeof(double) * 3 * (size_t)n), *i2 = (double *)malloc(sizeof(double) * 3 * (size_t)n);
  for (int64_t i = 0; i < n; ++i) {
    const int in = pnec_oracle_reprojection_score(bvs1 + 3 * i, bvs2 + 3 * i, best_R, best_t) < threshold;
    inlier_mask[i] = (uint8_t)in;
    if (in) {
      memcpy(i1 + 3 * cnt, bvs1 + 3 * i, 3 * sizeof(double));
      memcpy(i2 + 3 * cnt, bvs2 + 3 * i, 3 * sizeof(double));
      ++cnt;
    }
  }
  double v[3];
  pnec_oracle_rot_to_cayley(best_R, v);
  es_data D = {cnt, i1, i2};
  eigensolver_cayley(&D, v);
  pnec_oracle_cayley_to_rot(v, R_out);
  double M[9];
  pnec_oracle_compose_m(cnt, i1, i2, R_out, 1, M); /* the reference's ComposeM on the inliers (C7) */
  pnec_oracle_translation_from_m(M, t_out);
  free(i1);
  free(i2);
  *n_inliers = cnt;
  if (iterations) *iterations = it;
  return 0;
}

/* ---- PNEC::Solve with the reference's default Options, for a ragged batch: pnec.cc:77-124 ---------
 * ES_solution = Eigensolver(bvs1, bvs2, initial_pose, inliers) with use_ransac_ (pnec.cc:231-272)
 * InlierExtraction (pnec.cc:210-229)
 * weighted_iterations_ > 1: WeightedEigensolver on the inliers from ES_solution (pnec.cc:283-348)
 * use_ceres_: CeresSolver on the inliers from that (pnec.cc:350-370: default options, Target frame,
 * central numeric differences)
 * OpenMP over pairs (test tooling: the device's one-call chain against this at tens of thousands of
 * pairs).  Pair p draws its RANSAC samples as pair_id = first_pair_id + p.  Every intermediate the
 * parity tests look at comes back: the stage outputs, the mask, the iteration counts. */
void pnec_oracle_solve_chain_batch(int64_t n_pairs, const int64_t *offsets, const double *bvs1,
                                   const double *bvs2, const double *covs, const double *init_q,
                                   uint64_t seed, uint64_t first_pair_id, int max_ransac_iterations,
                                   int sample_size, double threshold, double reg, int weighted_iterations,
                                   int num_threads, double *es_q, double *es_t, uint8_t *inlier_mask,
                                   int32_t *inlier_count, int32_t *ransac_iterations, double *w_q,
                                   double *w_t, double *out_q, double *out_t, int32_t *ls_iterations,
                                   int32_t *ls_status) {
  pnec_oracle_options opt;
  pnec_oracle_default_options(&opt); /* PNEC::CeresSolver default-constructs its optimiser (pnec.cc:355) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t a = offsets[p], n = offsets[p + 1] - offsets[p];
    double R0[9], Rr[9], tr[3], Rw[9], tw[3], qw[4];
    pnec_oracle_rot_from_quat(init_q + 4 * p, R0);
    int32_t cnt = 0, its = 0;
    pnec_oracle_ransac_eigensolver(n, bvs1 + 3 * a, bvs2 + 3 * a, R0, seed, first_pair_id + (uint64_t)p,
                                   max_ransac_iterations, sample_size, threshold, Rr, tr, inlier_mask + a, &cnt,
                                   &its);
    inlier_count[p] = cnt;
    ransac_iterations[p] = its;
    pnec_oracle_quat_from_rot(Rr, es_q + 4 * p);
    memcpy(es_t + 3 * p, tr, sizeof(tr));
    const size_t m = (size_t)(cnt > 0 ? cnt : 1);
    double *i1 = (double *)malloc(sizeof(double) * 3 * m), *i2 = (double *)malloc(sizeof(double) * 3 * m);
    double *ic = (double *)malloc(sizeof(double) * 9 * m);
    int64_t k = 0;
    for (int64_t i = 0; i < n; ++i)
      if (inlier_mask[a + i]) {
        memcpy(i1 + 3 * k, bvs1 + 3 * (a + i), 3 * sizeof(double));
        memcpy(i2 + 3 * k, bvs2 + 3 * (a + i), 3 * sizeof(double));
        memcpy(ic + 9 * k, covs + 9 * (a + i), 9 * sizeof(double));
        ++k;
      }
    if (weighted_iterations > 1) {
      pnec_oracle_weighted_eigensolver_ex(k, i1, i2, ic, Rr, tr, reg, weighted_iterations, 0, Rw, tw);
    } else { /* == 1: the eigensolver's pose goes on (pnec.cc:109-111); 0 is not this function's business */
      memcpy(Rw, Rr, sizeof(Rw));
      memcpy(tw, tr, sizeof(tw));
    }
    pnec_oracle_quat_from_rot(Rw, qw);
    memcpy(w_q + 4 * p, qw, sizeof(qw));
    memcpy(w_t + 3 * p, tw, sizeof(tw));
    double cost = 0.0;
    int32_t it_ls = 0;
    const int st = pnec_oracle_solve(PNEC_ORACLE_MODE_TARGET, k, i1, i2, ic, NULL, reg, qw, tw, &opt, out_q + 4 * p,
                                     out_t + 3 * p, NULL, &cost, &it_ls);
    ls_iterations[p] = it_ls;
    ls_status[p] = st;
    free(i1);
    free(i2);
    free(ic);
  }
}
