/*
 * pnec_oracle_opengv.c -- TEST INFRASTRUCTURE ONLY (same rules as pnec_oracle.c).
 *
 * [EXT, FROM MEMORY, UNPINNED] Two restatements of what opengv::relative_pose::eigensolver may run as its
 * eigenvalue minimisation.  opengv is NOT in the reference tree (basalt master, un-pinned), the image
 * holds no copy and there is no network: nothing below can be checked against its source here.  What is written down
 * is a recollection of opengv's public master (src/relative_pose/methods.cpp, src/relative_pose/modules/main.cpp,
 * src/relative_pose/modules/eigensolver/modules.cpp, src/math/cayley.cpp), with the pieces the reference itself
 * corroborates marked as such.
 *
 *  What both schemes share (recalled, and consistent with the reference's own copy of the tail of that routine,
 *  pnec::common::TranslationFromM, src/common/common.cc:157-181 -- EigenSolver, the myPair sort, the
 *  "translationMagnitude * V.col(2)" line are opengv's eigensolver_main almost verbatim):
 *    relative_pose::eigensolver(adapter, indices, output)    builds the six 3x3 summation terms
 *        xxF = sum w f1x f1x (f2 f2'), yyF, zzF, xyF, yzF, zxF                      (= the 36 sums G_kl[a][c] here)
 *    and calls modules::eigensolver_main(xxF, ..., output), which starts at cayley = rot2cayley(output.rotation) and
 *    minimises the smallest eigenvalue of M(cayley) composed from those sums; M is composed with
 *    math::cayley2rot_REDUCED -- the rotation WITHOUT its 1 / (1 + |v|^2) scale -- so the function minimised is
 *        f(v) = lambda_min( M(N(v)) ) = (1 + |v|^2)^2 lambda_min( M(R(v)) ),   N(v) = (1 - |v|^2) I + 2 [v]x + 2 v v'.
 *    Its minimiser is NOT exactly that of lambda_min(M(R(v))): grad f = s^2 grad lambda + 4 s lambda v, so with noisy
 *    data (lambda_min > 0) it sits ~4 |v| lambda_min / (s |H|) away -- 1e-7..1e-6 rad on the benchmark's simulated
 *    pairs (|v| up to 0.4), 1e-9 on KITTI-like motion (|v| ~ 0.01).  (tests/test_opengv_schemes.py measures it.)
 *
 *  Scheme 2 -- "opengv-LM": what eigensolver_main does, as recalled:
 *        struct Eigensolver_step : OptimizationFunctor<double>   (values() = 3, inputs() = 3)
 *            operator()(x, fvec): getSmallestEVwithJacobian(xxF, ..., cayley = x, jacobian);  fvec = jacobian
 *        NumericalDiff<Eigensolver_step> numDiff(functor);                 (Forward, epsfcn = 0)
 *        LevenbergMarquardt<NumericalDiff<Eigensolver_step>> lm(numDiff);
 *        lm.resetParameters(); lm.parameters.ftol = 0.00005; lm.parameters.xtol = 1.E1 * epsilon;
 *        lm.parameters.maxfev = 100;  lm.minimize(x);
 *    i.e. Eigen's port of MINPACK lmder on the 3-vector F(v) = grad f(v) (the ANALYTIC gradient of the smallest root
 *    of M's characteristic polynomial), whose Jacobian -- the Hessian of f -- comes from forward differences
 *    (h_j = sqrt(eps) |x_j|, or sqrt(eps) when x_j = 0).  It looks for a ROOT OF THE GRADIENT: from a start in the
 *    basin it converges quadratically to the minimiser and then spends a few evaluations at the gradient's rounding
 *    floor until MINPACK's tests end it (with par = 0 the predicted reduction is 1, so ftol cannot end it before that).
 *    Restated below as MINPACK's algorithm (lmder's outer/inner loop and lmpar's iteration, Eigen's parameter
 *    defaults: factor 100, gtol 0), with the 3x3 linear algebra done on the normal equations instead of the pivoted
 *    QR factorisation -- the same step in exact arithmetic; Eigen's bits cannot be reproduced from memory anyway.
 *
 *  Scheme 1 -- "descent": the iteration VERDICT round 4 asked for (its judge, and the round-4 builder, remembered it
 *    as eigensolver_main): steepest descent along the NORMALISED gradient with step length lambda: start 0.01;
 *    in the first iteration doubled while the value keeps falling, up to 0.08; halved while a step does not improve
 *    the value; at most 50 iterations; stop once lambda < 1e-5.  As recalled HERE that loop -- with its constants
 *    lambda = 0.01, maxLambda = 0.08, modifier = 2.0, maxIterations = 50, min_xtol = 0.00001, disablingIncrements --
 *    is modules::ge_main2, the GENERALISED eigensolver (multi-camera; "this one doesn't work, probably because of double
 *    numerical differentiation -- use ge_main2, which is an implementation of gradient descent" sits above it), not the
 *    central one the reference calls (pnec.cc:236,274,310: CentralRelativeAdapter + relative_pose::eigensolver).  It
 *    is kept as a scheme because the two recollections disagree and neither can be checked: it stops ~lambda short
 *    of the minimiser (1e-5 rad), which is the one place where the choice is visible above the 1e-6 rad bar.
 *    ge_main2's "wrong minimum" branch (|cayley| < 0.01 and the second eigenvalue > 0.001 -> restart from a start
 *    disturbed by +-0.3, +-0.6 after three trials, at most five trials) is restated too, behind its own switch and OFF
 *    by default: for the central problem with unit bearings the second eigenvalue of M is 1e-4..5e-4 per
 *    correspondence (0.01..0.5 on KITTI-like pairs of 100..1000 points), so the test can never be satisfied: for EVERY small rotation -- all of KITTI -- all five trials are spent and the end of the last one, a
 *    descent started up to 0.6 away, is returned (five times the work for nothing; tests/test_opengv_schemes.py shows
 *    it); it only makes sense for the 4x4 generalised problem it was written for.
 *
 *  Recalled, and reproduced behind a switch since round 6 (pnec_oracle_set_ransac_chained_starts in
 *  pnec_oracle_frontend.c; the device: PNEC_HIP_RANSAC_CHAINED_STARTS; off by default -- the difference is inside the
 *  statistical noise of opengv's rand() draws): EigensolverSacProblem::getSelectedDistancesToModel writes the
 *  scored model into the adapter (_adapter.sett12 / setR12) before triangulating, so the NEXT hypothesis' start is
 *  the last scored model's rotation + jitter, not the initial one.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "pnec_oracle.h"

static inline double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
/* packed symmetric index of (a,c) in a 3x3: 00 01 02 11 12 22 */
static inline int s3(int a, int c) {
  if (a > c) { const int t = a; a = c; c = t; }
  return a * 3 - a * (a - 1) / 2 + (c - a);
}

/* The 36 sums G[6 s3(k,l) + s3(a,c)] = sum_i f2k f2l f1a f1c: opengv's xxF ... zxF (methods.cpp, "Fill summation terms"). */
void pnec_oracle_sums36(int64_t n, const double *b1, const double *b2, double G[36]) {
  memset(G, 0, 36 * sizeof(double));
  for (int64_t i = 0; i < n; ++i) {
    const double *f1 = b1 + 3 * i, *f2 = b2 + 3 * i;
    const double p[6] = {f2[0] * f2[0], f2[0] * f2[1], f2[0] * f2[2], f2[1] * f2[1], f2[1] * f2[2], f2[2] * f2[2]};
    const double q[6] = {f1[0] * f1[0], f1[0] * f1[1], f1[0] * f1[2], f1[1] * f1[1], f1[1] * f1[2], f1[2] * f1[2]};
    for (int kl = 0; kl < 6; ++kl)
      for (int ac = 0; ac < 6; ++ac) G[6 * kl + ac] += p[kl] * q[ac];
  }
}

/* N(v) (reduced != 0: math::cayley2rot_reduced) or R(v) = N(v) / (1 + |v|^2) (math::cayley2rot); row-major */
static void cayley_matrix(const double v[3], int reduced, double R[9]) {
  const double x = v[0], y = v[1], z = v[2];
  const double s = reduced ? 1.0 : 1.0 / (1.0 + x * x + y * y + z * z);
  R[0] = s * (1 + x * x - y * y - z * z); R[1] = s * 2 * (x * y - z); R[2] = s * 2 * (x * z + y);
  R[3] = s * 2 * (x * y + z); R[4] = s * (1 - x * x + y * y - z * z); R[5] = s * 2 * (y * z - x);
  R[6] = s * 2 * (x * z - y); R[7] = s * 2 * (y * z + x); R[8] = s * (1 - x * x - y * y + z * z);
}

/* M = sum_kl [r_k]x G_kl [r_l]x'  (r_k = column k of the matrix): eigensolver::composeM's sums, written with cross
 * products.  n = f1 x (R f2) = -sum_k f2k [r_k]x f1, so n n' = sum_kl f2k f2l [r_k]x f1 f1' [r_l]x'. */
static void compose_m_sums(const double G[36], const double R[9], double M[9]) {
  double r[3][3];
  for (int k = 0; k < 3; ++k) { r[k][0] = R[k]; r[k][1] = R[3 + k]; r[k][2] = R[6 + k]; }
  memset(M, 0, 9 * sizeof(double));
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 3; ++l) {
      const double *Gp = G + 6 * s3(k, l);
      double T[3][3]; /* T = [r_k]x G: column j = r_k x (column j of G) */
      for (int j = 0; j < 3; ++j) {
        const double gj[3] = {Gp[s3(0, j)], Gp[s3(1, j)], Gp[s3(2, j)]};
        double c[3];
        cross3(r[k], gj, c);
        T[0][j] = c[0]; T[1][j] = c[1]; T[2][j] = c[2];
      }
      for (int i = 0; i < 3; ++i) { /* row i of T [r_l]x' = r_l x (row i of T) */
        double c[3];
        cross3(r[l], T[i], c);
        M[3 * i] += c[0]; M[3 * i + 1] += c[1]; M[3 * i + 2] += c[2];
      }
    }
  M[1] = M[3] = 0.5 * (M[1] + M[3]);
  M[2] = M[6] = 0.5 * (M[2] + M[6]);
  M[5] = M[7] = 0.5 * (M[5] + M[7]);
}

/* lambda_min of M(v) from the sums; g (optional): its gradient w.r.t. v (e' dM e -- what
 * eigensolver::getSmallestEVwithJacobian delivers through the characteristic polynomial's closed form); e (optional):
 * the eigenvector; ev2 (optional): the second eigenvalue. */
double pnec_oracle_es_value_grad_sums(const double G[36], const double v[3], int reduced, double *g, double *e_out,
                                      double *ev2) {
  double R[9], M[9], w[3], V[9];
  cayley_matrix(v, reduced, R);
  compose_m_sums(G, R, M);
  pnec_oracle_sym_eig3(M, w, V);
  const double e[3] = {V[0], V[3], V[6]};
  if (e_out) { e_out[0] = e[0]; e_out[1] = e[1]; e_out[2] = e[2]; }
  if (ev2) *ev2 = w[1];
  if (!g) return w[0];
  double r[3][3], y[3][3], q[3][3];
  for (int k = 0; k < 3; ++k) { r[k][0] = R[k]; r[k][1] = R[3 + k]; r[k][2] = R[6 + k]; }
  for (int l = 0; l < 3; ++l) cross3(e, r[l], y[l]);
  for (int k = 0; k < 3; ++k) { /* q_k = (sum_l G_kl y_l) x e */
    double z[3] = {0, 0, 0};
    for (int l = 0; l < 3; ++l) {
      const double *Gp = G + 6 * s3(k, l);
      for (int a = 0; a < 3; ++a) z[a] += Gp[s3(a, 0)] * y[l][0] + Gp[s3(a, 1)] * y[l][1] + Gp[s3(a, 2)] * y[l][2];
    }
    cross3(z, e, q[k]);
  }
  /* d lambda / d v_j = 2 sum_k (d r_k / d v_j) . q_k;  dN_j = -2 v_j I + 2 [e_j]x + 2 (e_j v' + v e_j');
   * reduced: d r_k = column k of dN_j;  else (dN_j - 2 v_j R)[:, k] / s */
  const double s = 1.0 + dot3(v, v);
  for (int j = 0; j < 3; ++j) {
    double dN[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) dN[4 * i] = -2.0 * v[j];
    const int a = (j + 1) % 3, b = (j + 2) % 3;
    dN[3 * b + a] += 2.0;
    dN[3 * a + b] -= 2.0;
    for (int i = 0; i < 3; ++i) {
      dN[3 * j + i] += 2.0 * v[i];
      dN[3 * i + j] += 2.0 * v[i];
    }
    double acc = 0.0;
    for (int k = 0; k < 3; ++k)
      for (int rr = 0; rr < 3; ++rr) {
        const double d = reduced ? dN[3 * rr + k] : (dN[3 * rr + k] - 2.0 * v[j] * R[3 * rr + k]) / s;
        acc += d * q[k][rr];
      }
    g[j] = 2.0 * acc;
  }
  return w[0];
}

/* ---- scheme 1: the normalised descent with an adaptive step (header) -------------------------------------------------
 * evals (optional): evaluations counted as the device's quad spends them -- one trip = value + gradient at four step
 * lengths (the ladder of an iteration), the start's own evaluation included. */
int pnec_oracle_es_descent(const double G[36], double v[3], int *evals) {
  double lam = 0.01;
  const double max_lam = 0.08, mod = 2.0, min_xtol = 1e-5;
  double g[3];
  double ev = pnec_oracle_es_value_grad_sums(G, v, 0, g, NULL, NULL);
  int it = 0, trips = 1;
  for (; it < 50; ++it) {
    const double nrm = sqrt(dot3(g, g));
    if (!(nrm > 0.0)) break;
    const double d[3] = {g[0] / nrm, g[1] / nrm, g[2] / nrm};
    double sp[3] = {v[0] - lam * d[0], v[1] - lam * d[1], v[2] - lam * d[2]};
    double sev = pnec_oracle_es_value_grad_sums(G, sp, 0, NULL, NULL, NULL);
    int ladder = 1; /* step lengths tried in this iteration */
    if (it == 0) {
      while (sev < ev) {
        ev = sev;
        if (lam * mod > max_lam) break;
        lam *= mod;
        for (int k = 0; k < 3; ++k) sp[k] = v[k] - lam * d[k];
        sev = pnec_oracle_es_value_grad_sums(G, sp, 0, NULL, NULL, NULL);
        ++ladder;
      }
    }
    int halvings = 0;
    while (sev > ev && lam > 1e-12) {
      lam /= mod;
      for (int k = 0; k < 3; ++k) sp[k] = v[k] - lam * d[k];
      sev = pnec_oracle_es_value_grad_sums(G, sp, 0, NULL, NULL, NULL);
      ++halvings;
    }
    /* the device's quad tries four lengths per trip, each with its gradient (the winner's is the next iteration's):
     * the first iteration's doubled lengths lam0 {1, 2, 4, 8} are one trip (a halving that follows a doubling lands on a
     * length of that trip), halvings from there come four to a trip; later iterations: lam {1, 1/2, 1/4, 1/8}, then
     * {1/16, ...} */
    trips += (it == 0) ? 1 + (ladder > 1 ? 0 : (halvings + 3) / 4) : halvings / 4 + 1;
    for (int k = 0; k < 3; ++k) v[k] = sp[k];
    ev = pnec_oracle_es_value_grad_sums(G, v, 0, g, NULL, NULL);
    if (lam < min_xtol) { ++it; break; }
  }
  if (evals) *evals = trips;
  return it;
}

/* ge_main2's outer loop around the descent [EXT]: |cayley| < 0.01 and the second eigenvalue > 0.001 -> another trial
 * from the start disturbed by +-0.3 (+-0.6 from the fourth trial on), five trials at most; the draws come from the
 * counter hash (seed, pair = stream, hypothesis = trial, draw = component).  OFF unless switched on (header: why). */
static int g_es_restart = 0;
void pnec_oracle_set_eigensolver_restart(int on) { g_es_restart = on; }
int pnec_oracle_get_eigensolver_restart(void) { return g_es_restart; }
int pnec_oracle_es_descent_restarts(const double G[36], double v[3], uint64_t seed, uint64_t stream, int *trials_out) {
  const double v_start[3] = {v[0], v[1], v[2]};
  double amp = 0.3;
  int trials = 0, it = 0, found = 0;
  while (!found && trials < 5) {
    if (trials > 2) amp = 0.6;
    for (int k = 0; k < 3; ++k)
      v[k] = v_start[k] + (trials == 0 ? 0.0 : (pnec_oracle_rng_uniform(seed, stream, (uint64_t)trials, 2000 + k) - 0.5) * 2.0 * amp);
    it = pnec_oracle_es_descent(G, v, NULL);
    if (sqrt(dot3(v, v)) < 0.01) {
      double ev2 = 0.0;
      pnec_oracle_es_value_grad_sums(G, v, 0, NULL, NULL, &ev2);
      if (ev2 > 0.001) ++trials;
      else found = 1;
    } else {
      found = 1;
    }
  }
  if (trials_out) *trials_out = trials;
  return it;
}

/* ---- scheme 2: Eigen's LevenbergMarquardt (MINPACK lmder/lmpar) on F(v) = grad f(v), f with the REDUCED rotation ---- */
static void es_F(const double G[36], const double x[3], double F[3]) { pnec_oracle_es_value_grad_sums(G, x, 1, F, NULL, NULL); }
static double nrm3(const double a[3]) { return sqrt(dot3(a, a)); }

/* Cholesky of a symmetric positive definite 3x3 (row-major, lower triangle read); 0 = not positive definite */
static int chol3(const double A[9], double L[6] /* l00 l10 l11 l20 l21 l22 */) {
  if (!(A[0] > 0.0)) return 0;
  L[0] = sqrt(A[0]);
  L[1] = A[3] / L[0];
  const double d1 = A[4] - L[1] * L[1];
  if (!(d1 > 0.0)) return 0;
  L[2] = sqrt(d1);
  L[3] = A[6] / L[0];
  L[4] = (A[7] - L[3] * L[1]) / L[2];
  const double d2 = A[8] - L[3] * L[3] - L[4] * L[4];
  if (!(d2 > 0.0)) return 0;
  L[5] = sqrt(d2);
  return 1;
}
static void chol3_forward(const double L[6], const double b[3], double z[3]) { /* L z = b */
  z[0] = b[0] / L[0];
  z[1] = (b[1] - L[1] * z[0]) / L[2];
  z[2] = (b[2] - L[3] * z[0] - L[4] * z[1]) / L[5];
}
static void chol3_backward(const double L[6], const double z[3], double x[3]) { /* L' x = z */
  x[2] = z[2] / L[5];
  x[1] = (z[1] - L[4] * x[2]) / L[2];
  x[0] = (z[0] - L[1] * x[1] - L[3] * x[2]) / L[0];
}

/* MINPACK lmpar on the normal equations: A = J'J, b = J'f.  On return x solves (A + par D^2) x = b (the caller steps
 * by -x) with |D x| within 10 % of delta, or par = 0 and x the Gauss-Newton solution when that lies inside. */
static void lmpar3(const double A[9], const double b[3], const double diag[3], double delta, double *par, double x[3]) {
  const double dwarf = DBL_MIN;
  double L[6], z[3], wa1[3], wa2[3];
  const int full_rank = chol3(A, L);
  double dxnorm, fp, parl = 0.0, paru, gnorm, temp;
  if (full_rank) {
    chol3_forward(L, b, z);
    chol3_backward(L, z, x);
  } else {
    x[0] = x[1] = x[2] = 0.0; /* rank-deficient Jacobian: MINPACK takes a basic least-squares solution; here none (the
                                 damped system below decides) */
  }
  for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
  dxnorm = nrm3(wa2);
  fp = dxnorm - delta;
  if (full_rank && fp <= 0.1 * delta) { *par = 0.0; return; }
  if (full_rank) { /* the Newton step gives a lower bound parl of the zero of the function */
    for (int j = 0; j < 3; ++j) wa1[j] = diag[j] * (wa2[j] / dxnorm);
    chol3_forward(L, wa1, z);
    temp = nrm3(z);
    parl = fp / delta / temp / temp;
  } else {
    fp = delta; /* (any positive value: the Gauss-Newton step is "outside") */
  }
  for (int j = 0; j < 3; ++j) wa1[j] = b[j] / diag[j];
  gnorm = nrm3(wa1);
  paru = gnorm / delta;
  if (paru == 0.0) paru = dwarf / fmin(delta, 0.1);
  *par = fmax(*par, parl);
  *par = fmin(*par, paru);
  if (*par == 0.0) *par = gnorm / dxnorm;
  for (int iter = 1;; ++iter) {
    if (*par == 0.0) *par = fmax(dwarf, 0.001 * paru);
    double Ap[9], Lp[6];
    memcpy(Ap, A, sizeof(Ap));
    for (int j = 0; j < 3; ++j) Ap[4 * j] += *par * diag[j] * diag[j];
    if (!chol3(Ap, Lp)) { /* cannot happen for par > 0 short of NaN / overflow */
      x[0] = x[1] = x[2] = 0.0;
      return;
    }
    chol3_forward(Lp, b, z);
    chol3_backward(Lp, z, x);
    for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = nrm3(wa2);
    temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
    for (int j = 0; j < 3; ++j) wa1[j] = diag[j] * (wa2[j] / dxnorm);
    chol3_forward(Lp, wa1, z);
    temp = nrm3(z);
    const double parc = fp / delta / temp / temp;
    if (fp > 0.0) parl = fmax(parl, *par);
    if (fp < 0.0) paru = fmin(paru, *par);
    *par = fmax(parl, *par + parc);
  }
}

/* info: Eigen's LevenbergMarquardtSpace::Status -- 1 RelativeReductionTooSmall, 2 RelativeErrorTooSmall, 3 both,
 * 4 CosinusTooSmall, 5 TooManyFunctionEvaluation, 6 FtolTooSmall, 7 XtolTooSmall, 8 GtolTooSmall.
 * Returns the number of successful iterations; nfev as Eigen counts it (1 + per outer iteration 4 + 1 per trial). */
int pnec_oracle_es_lm(const double G[36], double x[3], int *nfev_out, int *info_out) {
  const double ftol = 0.00005, xtol = 10.0 * DBL_EPSILON, gtol = 0.0, factor = 100.0;
  const int maxfev = 100;
  const double eps = sqrt(DBL_EPSILON);
  double fvec[3], J[9] /* row-major: J[3 r + c] = dF_r / dx_c */, diag[3] = {1, 1, 1}, wa2[3];
  int nfev = 1, iter = 1, info = 0;
  es_F(G, x, fvec);
  double fnorm = nrm3(fvec), par = 0.0, xnorm = 0.0, delta = 0.0;
  for (;;) {
    /* NumericalDiff<..., Forward>::df */
    for (int j = 0; j < 3; ++j) {
      double h = eps * fabs(x[j]);
      if (h == 0.0) h = eps;
      double xp[3] = {x[0], x[1], x[2]}, val2[3];
      xp[j] += h;
      es_F(G, xp, val2);
      for (int r = 0; r < 3; ++r) J[3 * r + j] = (val2[r] - fvec[r]) / h;
    }
    nfev += 4;
    for (int j = 0; j < 3; ++j) wa2[j] = sqrt(J[j] * J[j] + J[3 + j] * J[3 + j] + J[6 + j] * J[6 + j]);
    if (iter == 1) {
      for (int j = 0; j < 3; ++j) diag[j] = (wa2[j] == 0.0) ? 1.0 : wa2[j];
      const double dx[3] = {diag[0] * x[0], diag[1] * x[1], diag[2] * x[2]};
      xnorm = nrm3(dx);
      delta = factor * xnorm;
      if (delta == 0.0) delta = factor;
    }
    double A[9], b[3]; /* J'J, J'f */
    for (int r = 0; r < 3; ++r) {
      b[r] = J[r] * fvec[0] + J[3 + r] * fvec[1] + J[6 + r] * fvec[2];
      for (int c = 0; c < 3; ++c) A[3 * r + c] = J[r] * J[c] + J[3 + r] * J[3 + c] + J[6 + r] * J[6 + c];
    }
    double gnorm = 0.0;
    if (fnorm != 0.0)
      for (int j = 0; j < 3; ++j)
        if (wa2[j] != 0.0) gnorm = fmax(gnorm, fabs(b[j] / fnorm / wa2[j]));
    if (gnorm <= gtol) { info = 4; break; }
    for (int j = 0; j < 3; ++j) diag[j] = fmax(diag[j], wa2[j]);
    double ratio = 0.0;
    do {
      double p[3], xn[3], f1[3];
      lmpar3(A, b, diag, delta, &par, p);
      for (int j = 0; j < 3; ++j) { p[j] = -p[j]; xn[j] = x[j] + p[j]; }
      const double dp[3] = {diag[0] * p[0], diag[1] * p[1], diag[2] * p[2]};
      const double pnorm = nrm3(dp);
      if (iter == 1) delta = fmin(delta, pnorm);
      es_F(G, xn, f1);
      ++nfev;
      const double fnorm1 = nrm3(f1);
      double actred = -1.0;
      if (0.1 * fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
      const double Jp[3] = {J[0] * p[0] + J[1] * p[1] + J[2] * p[2], J[3] * p[0] + J[4] * p[1] + J[5] * p[2],
                            J[6] * p[0] + J[7] * p[1] + J[8] * p[2]};
      const double t1 = nrm3(Jp) / fnorm, t2 = sqrt(par) * pnorm / fnorm;
      const double temp1 = t1 * t1, temp2 = t2 * t2;
      const double prered = temp1 + temp2 / 0.5, dirder = -(temp1 + temp2);
      ratio = (prered != 0.0) ? actred / prered : 0.0;
      if (ratio <= 0.25) {
        double temp = 0.5;
        if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
        delta = temp * fmin(delta, pnorm / 0.1);
        par /= temp;
      } else if (!(par != 0.0 && ratio < 0.75)) {
        delta = pnorm / 0.5;
        par = 0.5 * par;
      }
      if (ratio >= 1e-4) {
        for (int j = 0; j < 3; ++j) { x[j] = xn[j]; fvec[j] = f1[j]; }
        const double dx[3] = {diag[0] * x[0], diag[1] * x[1], diag[2] * x[2]};
        xnorm = nrm3(dx);
        fnorm = fnorm1;
        ++iter;
      }
      const int small_red = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0;
      if (small_red && delta <= xtol * xnorm) { info = 3; break; }
      if (small_red) { info = 1; break; }
      if (delta <= xtol * xnorm) { info = 2; break; }
      if (nfev >= maxfev) { info = 5; break; }
      if (fabs(actred) <= DBL_EPSILON && prered <= DBL_EPSILON && 0.5 * ratio <= 1.0) { info = 6; break; }
      if (delta <= DBL_EPSILON * xnorm) { info = 7; break; }
      if (gnorm <= DBL_EPSILON) { info = 8; break; }
    } while (ratio < 1e-4);
    if (info) break;
  }
  if (nfev_out) *nfev_out = nfev;
  if (info_out) *info_out = info;
  return iter - 1;
}
