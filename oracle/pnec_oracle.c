/*
 * pnec_oracle.c -- TEST INFRASTRUCTURE ONLY (see pnec_oracle.h for scope and parity status).
 *
 * Plain-C restatement of the reference's PNEC / NEC least-squares refinement:
 *   residual functors        include/optimization/pnec_residual.h:50-150, nec_residual.h:47-68
 *   problem construction     src/optimization/pnec_ceres.cc:70-168 (one 1-D residual block per
 *                            correspondence over theta(1), phi(1), quaternion(4); central numeric
 *                            differences; EigenQuaternionManifold on the quaternion; no loss)
 *   result                   src/optimization/pnec_ceres.cc:192-207
 *   driver                   src/rel_pose_estimation/pnec.cc:350-411
 * The minimiser is Ceres (not in the tree, version unpinned): the trust-region /
 * Levenberg-Marquardt loop below restates Ceres 2.1's published algorithm with default
 * ceres::Solver::Options.  "parity unpinned" for that part.
 */
#include "pnec_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ small vector helpers */
static inline double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
static inline void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
/* y = R x, R row-major */
static inline void mat_vec(const double R[9], const double x[3], double y[3]) {
  y[0] = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  y[1] = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  y[2] = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
}
/* y = R^T x */
static inline void matT_vec(const double R[9], const double x[3], double y[3]) {
  y[0] = R[0] * x[0] + R[3] * x[1] + R[6] * x[2];
  y[1] = R[1] * x[0] + R[4] * x[1] + R[7] * x[2];
  y[2] = R[2] * x[0] + R[5] * x[1] + R[8] * x[2];
}
/* common.cc:96-101, row-major output */
static inline void skew(const double v[3], double S[9]) {
  S[0] = 0.0;   S[1] = -v[2]; S[2] = v[1];
  S[3] = v[2];  S[4] = 0.0;   S[5] = -v[0];
  S[6] = -v[1]; S[7] = v[0];  S[8] = 0.0;
}
/* v' C v with C an Eigen column-major 3x3: C(r,c) = C[3*c + r] */
static inline double quad_form_colmajor(const double *C, const double v[3]) {
  double acc = 0.0;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) acc += v[r] * C[3 * c + r] * v[c];
  return acc;
}
/* y = 1/2 (C + C') v, C column-major */
static inline void sym_mat_vec_colmajor(const double *C, const double v[3], double y[3]) {
  for (int r = 0; r < 3; ++r) {
    double a = 0.0;
    for (int c = 0; c < 3; ++c) a += 0.5 * (C[3 * c + r] + C[3 * r + c]) * v[c];
    y[r] = a;
  }
}

/* ------------------------------------------------------------------ exported small pieces */
void pnec_oracle_default_options(pnec_oracle_options *o) {
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->check_convergence = 1;
  o->jacobian_mode = PNEC_ORACLE_JAC_NUMERIC_CENTRAL;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
}

/* common.cc:103-116 (abs() there resolves to the double overload: <math.h> is included) */
void pnec_oracle_angles_from_vec(const double v[3], double *theta, double *phi) {
  const double n = sqrt(dot3(v, v));
  if (n == 0.0) {
    *theta = 0.0;
    *phi = 0.0;
    return;
  }
  const double x = v[0] / n, y = v[1] / n, z = v[2] / n;
  *theta = acos(z);
  if (fabs(*theta) < 1e-10) {
    *phi = 0.0;
  } else {
    *phi = atan2(y, x);
  }
}

/* Eigen::Quaterniond(const Matrix3d&) [EXT: Eigen/src/Geometry/Quaternion.h, Shepperd's method] */
void pnec_oracle_quat_from_rot(const double R[9], double q[4]) {
  double tr = R[0] + R[4] + R[8];
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
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
}

/* Eigen::QuaternionBase::toRotationMatrix() -- no normalisation (pnec_residual.h:92-93) */
void pnec_oracle_rot_from_quat(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

static inline void translation_from_angles(double theta, double phi, double t[3]) {
  t[0] = sin(theta) * cos(phi);
  t[1] = sin(theta) * sin(phi);
  t[2] = cos(theta);
}

/* pnec_ceres.cc:201-207 */
void pnec_oracle_result(const double q[4], double theta, double phi, double R[9], double t[3]) {
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double qn[4] = {q[0] / n, q[1] / n, q[2] / n, q[3] / n};
  pnec_oracle_rot_from_quat(qn, R);
  translation_from_angles(theta, phi, t);
}

/* common.cc:210-214; Sophus::SO3d::logAndTheta on the unit quaternion of R1^T R2 [EXT] */
double pnec_oracle_rotational_difference_deg(const double R1[9], const double R2[9]) {
  double D[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += R1[3 * k + i] * R2[3 * k + j];
      D[3 * i + j] = a;
    }
  double q[4];
  pnec_oracle_quat_from_rot(D, q);
  const double qn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[3] / qn;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) / qn;
  const double theta = (w < 0.0) ? 2.0 * atan2(-n, -w) : 2.0 * atan2(n, w);
  return fabs(theta) * 180.0 / M_PI;
}

/* common.cc:216-235 (including the duplicated translation_1 test at :220) */
double pnec_oracle_translational_difference_deg(const double t1[3], const double t2[3],
                                                int both_directions) {
  const double n1 = sqrt(dot3(t1, t1)), n2 = sqrt(dot3(t2, t2));
  double error;
  if (n1 < 1e-10 || n1 < 1e-10) {
    error = M_PI / 2.0;
  } else if (both_directions) {
    const double e1 = acos(dot3(t1, t2) / (n1 * n2));
    const double e2 = acos(-dot3(t1, t2) / (n1 * n2));
    error = e1 < e2 ? e1 : e2;
  } else {
    error = acos(dot3(t1, t2) / (n1 * n2));
  }
  return error * 180.0 / M_PI;
}

/* common.cc:237-259 */
double pnec_oracle_cost_function(int64_t n, const double *bvs1, const double *bvs2,
                                 const double *covs, const double R[9], const double t[3]) {
  double cost = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double *f1 = bvs1 + 3 * i, *f2 = bvs2 + 3 * i, *C = covs + 9 * i;
    double F[9], a[3], v[3], Rf2[3], c[3];
    skew(f1, F);
    matT_vec(F, t, a);  /* (t' f1hat)' */
    matT_vec(R, a, v);  /* (t' f1hat R)' */
    mat_vec(R, f2, Rf2);
    cross3(f1, Rf2, c);
    const double num = dot3(t, c);
    cost += num * num / quad_form_colmajor(C, v);
  }
  return cost / (double)n;
}

/* ------------------------------------------------------------------ covariance propagation */
/* pnec::common::RotationBetweenPoints(point1, point2) (common.cc:118-124): I + [v]x + [v]x^2 / (1 + c), v = p1 x p2,
 * c = p1.p2, for unit vectors (the C++ does not normalise; scripts/pnec/math.py:42-64 does, which is the same thing for
 * unit input).  Column-major output, like every 3x3 this file hands out. */
void pnec_oracle_rotation_between_points(const double p1[3], const double p2[3], double Rm[9]) {
  double c[3], K[9], K2[9];
  cross3(p1, p2, c);
  skew(c, K); /* row-major */
  const double d = dot3(p1, p2);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = a;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rm[3 * j + i] = (i == j ? 1.0 : 0.0) + K[3 * i + j] + K2[3 * i + j] / (1.0 + d);
}

/* common.cc:118-124 with point1 = (0,0,1); column-major output */
static void rotation_from_z(const double v[3], double Rm[9]) {
  const double z[3] = {0.0, 0.0, 1.0};
  double c[3], K[9], K2[9];
  cross3(z, v, c);
  skew(c, K); /* row-major */
  const double d = v[2];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0;
      for (int k = 0; k < 3; ++k) a += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = a;
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      Rm[3 * j + i] = (i == j ? 1.0 : 0.0) + K[3 * i + j] + K2[3 * i + j] / (1.0 + d);
}

void pnec_oracle_unproject(const double img_pt[2], const double K_inv[9], double out[3]) {
  const double mu[3] = {img_pt[0], img_pt[1], 1.0};
  double n = 0.0;
  for (int r = 0; r < 3; ++r) {
    out[r] = K_inv[r] * mu[0] + K_inv[3 + r] * mu[1] + K_inv[6 + r] * mu[2];
    n += out[r] * out[r];
  }
  n = sqrt(n);
  for (int r = 0; r < 3; ++r) out[r] /= n;
}

void pnec_oracle_unscented_transform(const double mu[3], const double cov[9], const double K_inv[9],
                                     double kappa, int camera_model, double out[9]) {
  const int n = 2, m = 5;
  double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* column-major */
  double Rm[9];
  double a, b, d;
  if (camera_model == 0) { /* Omnidirectional */
    const double nm = sqrt(dot3(mu, mu));
    const double v[3] = {mu[0] / nm, mu[1] / nm, mu[2] / nm};
    rotation_from_z(v, Rm);
    /* local = R' cov R, top-left 2x2 */
    double T[9], Lc[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += cov[3 * k + i] * Rm[3 * j + k]; /* (cov R)(i,j) */
        T[3 * j + i] = s;
      }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += Rm[3 * i + k] * T[3 * j + k]; /* R'(i,k) = R(k,i) */
        Lc[3 * j + i] = s;
      }
    a = Lc[0]; b = Lc[1]; d = Lc[4];
  } else {
    a = cov[0]; b = cov[1]; d = cov[4];
  }
  const double l00 = sqrt(a), l10 = b / l00, l11 = sqrt(d - l10 * l10);
  C[0] = l00; C[1] = l10; C[4] = l11; /* lower factor: (0,0), (1,0), (1,1) */
  if (camera_model == 0) {
    double RC[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += Rm[3 * k + i] * C[3 * j + k];
        RC[3 * j + i] = s;
      }
    memcpy(C, RC, sizeof(C));
  }
  double pts[5][3], w[5], tp[5][3], mean[3] = {0, 0, 0};
  w[0] = kappa / ((double)n + kappa);
  for (int k = 0; k < 3; ++k) pts[0][k] = mu[k];
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 3; ++k) {
      pts[1 + i][k] = mu[k] + C[3 * i + k];
      pts[1 + n + i][k] = mu[k] - C[3 * i + k];
    }
    w[1 + i] = w[1 + n + i] = 0.5 / ((double)n + kappa);
  }
  for (int i = 0; i < m; ++i) {
    double t[3];
    if (camera_model == 0) {
      memcpy(t, pts[i], sizeof(t));
    } else {
      for (int r = 0; r < 3; ++r)
        t[r] = K_inv[r] * pts[i][0] + K_inv[3 + r] * pts[i][1] + K_inv[6 + r] * pts[i][2];
    }
    const double nt = sqrt(dot3(t, t));
    for (int k = 0; k < 3; ++k) {
      tp[i][k] = t[k] / nt;
      mean[k] += w[i] * tp[i][k];
    }
  }
  memset(out, 0, 9 * sizeof(double));
  for (int i = 0; i < m; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) out[3 * c + r] += w[i] * (tp[i][r] - mean[r]) * (tp[i][c] - mean[c]);
}

/* ------------------------------------------------------------------ residuals (literal) */
static double residual_core(int mode, const double f1[3], const double f2[3], const double *cov2,
                            const double *cov1, double reg, const double R[9],
                            const double t[3]) {
  double Rf2[3], c[3];
  mat_vec(R, f2, Rf2);
  cross3(f1, Rf2, c);
  const double num = dot3(t, c); /* t' (f1 x R f2) */
  if (mode == PNEC_ORACLE_MODE_NEC) return num;

  double den = reg;
  if (mode == PNEC_ORACLE_MODE_TARGET || mode == PNEC_ORACLE_MODE_SYM) {
    /* t' f1hat R Sigma R' f1hat' t  (pnec_residual.h:97-102 / :133-136) */
    double F[9], a[3], v[3];
    skew(f1, F);
    matT_vec(F, t, a);
    matT_vec(R, a, v);
    den += quad_form_colmajor(cov2, v);
  }
  if (mode == PNEC_ORACLE_MODE_HOST) {
    /* t' [R f1]x Sigma [R f1]x' t  (pnec_residual.h:63-70); the single cov array is in frame 1 */
    double p[3], A[9], v[3];
    mat_vec(R, f1, p);
    skew(p, A);
    matT_vec(A, t, v);
    den += quad_form_colmajor(cov2, v);
  }
  if (mode == PNEC_ORACLE_MODE_SYM) {
    /* + t' [R f2]x Sigma_1 [R f2]x' t  (pnec_residual.h:137-138) */
    double A[9], v[3];
    skew(Rf2, A);
    matT_vec(A, t, v);
    den += quad_form_colmajor(cov1, v);
  }
  return num / sqrt(den);
}

double pnec_oracle_residual(int mode, const double f1[3], const double f2[3], const double *cov2,
                            const double *cov1, double reg, double theta, double phi,
                            const double q[4]) {
  /* the functors rebuild t and R on every call; so do we (keeps the CPU timing honest) */
  double t[3], R[9];
  translation_from_angles(theta, phi, t);
  pnec_oracle_rot_from_quat(q, R);
  return residual_core(mode, f1, f2, cov2, cov1, reg, R, t);
}

double pnec_oracle_energy(int mode, int64_t n, const double *bvs1, const double *bvs2,
                          const double *covs2, const double *covs1, double reg,
                          const double R[9], const double t[3]) {
  double e = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double r = residual_core(mode, bvs1 + 3 * i, bvs2 + 3 * i,
                                   covs2 ? covs2 + 9 * i : NULL, covs1 ? covs1 + 9 * i : NULL,
                                   reg, R, t);
    e += r * r;
  }
  return e;
}

/* ------------------------------------------------------------------ Jacobians */
/* EigenQuaternionManifold::PlusJacobian [EXT], 4x3 row-major, storage xyzw */
static void quat_plus_jacobian(const double q[4], double P[12]) {
  P[0] = q[3];  P[1] = q[2];   P[2] = -q[1];
  P[3] = -q[2]; P[4] = q[3];   P[5] = q[0];
  P[6] = q[1];  P[7] = -q[0];  P[8] = q[3];
  P[9] = -q[0]; P[10] = -q[1]; P[11] = -q[2];
}

static double g_numeric_step_scale = 1.0;
void pnec_oracle_set_numeric_step_scale(double s) { g_numeric_step_scale = s > 0.0 ? s : 1.0; }

/* NumericDiffCostFunction<..., CENTRAL, 1, 1, 1, 4> [EXT]: per ambient parameter
 * h = max(sqrt(eps), 1e-6 |x_j|), J_j = (f(x + h e_j) - f(x - h e_j)) / (2h); the quaternion is
 * perturbed component-wise with no renormalisation; the manifold's plus-Jacobian then maps
 * the 1x4 block to the 3-dim tangent space. */
static void numeric_row(int mode, const double f1[3], const double f2[3], const double *cov2,
                        const double *cov1, double reg, double theta, double phi,
                        const double q[4], const double P[12], double *r, double J[5]) {
  /* g_numeric_step_scale (1.0; test tooling): the same derivative through a slightly different step -- what the result
   * owes to the ROUNDING of the difference quotient rather than to the derivative it approximates */
  const double kMinStep = 1.4901161193847656e-08 * g_numeric_step_scale; /* sqrt(DBL_EPSILON) */
  const double kRel = 1e-6 * g_numeric_step_scale;
  *r = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta, phi, q);
  {
    const double h = fmax(kMinStep, fabs(theta) * kRel);
    const double rp = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta + h, phi, q);
    const double rm = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta - h, phi, q);
    J[0] = (rp - rm) * (1.0 / h / 2.0);
  }
  {
    const double h = fmax(kMinStep, fabs(phi) * kRel);
    const double rp = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta, phi + h, q);
    const double rm = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta, phi - h, q);
    J[1] = (rp - rm) * (1.0 / h / 2.0);
  }
  double Jq[4];
  for (int j = 0; j < 4; ++j) {
    const double h = fmax(kMinStep, fabs(q[j]) * kRel);
    double qp[4] = {q[0], q[1], q[2], q[3]};
    qp[j] = q[j] + h;
    const double rp = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta, phi, qp);
    qp[j] = q[j] - h;
    const double rm = pnec_oracle_residual(mode, f1, f2, cov2, cov1, reg, theta, phi, qp);
    Jq[j] = (rp - rm) * (1.0 / h / 2.0);
  }
  for (int c = 0; c < 3; ++c)
    J[2 + c] = Jq[0] * P[c] + Jq[1] * P[3 + c] + Jq[2] * P[6 + c] + Jq[3] * P[9 + c];
}

/* Closed form.  Left perturbation R <- Exp(omega) R, omega = 2 delta
 * (EigenQuaternionManifold::Plus is q_delta * q with a half-angle vector); t <- t + B eta with
 * B = d t / d(theta, phi). */
typedef struct {
  double R[9], t[3], Bth[3], Bph[3];
} pose_uniforms;

static void make_uniforms(double theta, double phi, const double q[4], pose_uniforms *u) {
  pnec_oracle_rot_from_quat(q, u->R);
  const double st = sin(theta), ct = cos(theta), sp = sin(phi), cp = cos(phi);
  u->t[0] = st * cp;   u->t[1] = st * sp;   u->t[2] = ct;
  u->Bth[0] = ct * cp; u->Bth[1] = ct * sp; u->Bth[2] = -st;
  u->Bph[0] = -st * sp; u->Bph[1] = st * cp; u->Bph[2] = 0.0;
}

static void analytic_row(int mode, const double f1[3], const double f2[3], const double *cov2,
                         const double *cov1, double reg, const pose_uniforms *U, double *r,
                         double J[5]) {
  const double *R = U->R, *t = U->t;
  double m[3], g[3];
  cross3(t, f1, m);   /* m = t x f1 = f1hat' t */
  matT_vec(R, m, g);  /* g = R' m */
  const double n = dot3(f2, g);
  double Jw[3], Jt[3];
  if (mode == PNEC_ORACLE_MODE_NEC) {
    double u[3];
    mat_vec(R, f2, u);
    *r = n;
    cross3(u, m, Jw);
    cross3(f1, u, Jt);
  } else if (mode == PNEC_ORACLE_MODE_TARGET) {
    double Sg[3], w[3], u[3];
    sym_mat_vec_colmajor(cov2, g, Sg);
    const double d = dot3(g, Sg) + reg, s = sqrt(d);
    *r = n / s;
    for (int k = 0; k < 3; ++k) w[k] = f2[k] / s - n * Sg[k] / (s * d);
    mat_vec(R, w, u);
    cross3(u, m, Jw);
    cross3(f1, u, Jt);
  } else {
    /* HOST: d = h' S h + reg, h = t x (R f1).  SYM: d = g' S2 g + h' S1 h + reg, h = t x (R f2) */
    const int sym = (mode == PNEC_ORACLE_MODE_SYM);
    double p[3], h[3], Sh[3], Sg[3] = {0, 0, 0};
    mat_vec(R, sym ? f2 : f1, p);
    cross3(t, p, h);
    sym_mat_vec_colmajor(sym ? cov1 : cov2, h, Sh);
    double d = dot3(h, Sh) + reg;
    if (sym) {
      sym_mat_vec_colmajor(cov2, g, Sg);
      d += dot3(g, Sg);
    }
    const double s = sqrt(d);
    *r = n / s;
    double w[3], u[3], wh[3], a[3], b[3], c[3];
    for (int k = 0; k < 3; ++k) {
      w[k] = f2[k] / s - n * Sg[k] / (s * d);
      wh[k] = -n * Sh[k] / (s * d);
    }
    mat_vec(R, w, u);
    cross3(u, m, Jw);
    cross3(f1, u, Jt);
    cross3(wh, t, a);
    cross3(p, a, b); /* p x (wh x t) */
    cross3(p, wh, c);
    for (int k = 0; k < 3; ++k) {
      Jw[k] += b[k];
      Jt[k] += c[k];
    }
  }
  J[0] = dot3(U->Bth, Jt);
  J[1] = dot3(U->Bph, Jt);
  J[2] = 2.0 * Jw[0];
  J[3] = 2.0 * Jw[1];
  J[4] = 2.0 * Jw[2];
}

void pnec_oracle_evaluate(int mode, int jacobian_mode, int64_t n, const double *bvs1,
                          const double *bvs2, const double *covs2, const double *covs1,
                          double reg, double theta, double phi, const double q[4], double *r,
                          double *J, double *cost) {
  double P[12];
  pose_uniforms U;
  quat_plus_jacobian(q, P);
  make_uniforms(theta, phi, q, &U);
  double c = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double *f1 = bvs1 + 3 * i, *f2 = bvs2 + 3 * i;
    const double *c2 = covs2 ? covs2 + 9 * i : NULL, *c1 = covs1 ? covs1 + 9 * i : NULL;
    double ri, Ji[5];
    if (jacobian_mode == PNEC_ORACLE_JAC_ANALYTIC)
      analytic_row(mode, f1, f2, c2, c1, reg, &U, &ri, Ji);
    else
      numeric_row(mode, f1, f2, c2, c1, reg, theta, phi, q, P, &ri, Ji);
    if (r) r[i] = ri;
    if (J) memcpy(J + 5 * i, Ji, sizeof(Ji));
    c += ri * ri;
  }
  if (cost) *cost = 0.5 * c;
}

/* ------------------------------------------------------------------ the minimiser */
typedef struct {
  int mode, jac_mode;
  int64_t n;
  const double *b1, *b2, *c2, *c1;
  double reg;
} problem_t;

typedef struct {
  double theta, phi, q[4];
} state_t;

static double state_norm(const state_t *x) {
  return sqrt(x->theta * x->theta + x->phi * x->phi + x->q[0] * x->q[0] + x->q[1] * x->q[1] +
              x->q[2] * x->q[2] + x->q[3] * x->q[3]);
}
static double state_dist(const state_t *a, const state_t *b) {
  double s = (a->theta - b->theta) * (a->theta - b->theta) + (a->phi - b->phi) * (a->phi - b->phi);
  for (int k = 0; k < 4; ++k) s += (a->q[k] - b->q[k]) * (a->q[k] - b->q[k]);
  return sqrt(s);
}

/* cost = 1/2 sum r^2 only (Ceres evaluates the candidate without Jacobians) */
static int eval_cost(const problem_t *P, const state_t *x, double *cost) {
  double c = 0.0;
  if (P->jac_mode == PNEC_ORACLE_JAC_ANALYTIC) {
    double R[9], t[3];
    translation_from_angles(x->theta, x->phi, t);
    pnec_oracle_rot_from_quat(x->q, R);
    for (int64_t i = 0; i < P->n; ++i) {
      const double r = residual_core(P->mode, P->b1 + 3 * i, P->b2 + 3 * i,
                                     P->c2 ? P->c2 + 9 * i : NULL, P->c1 ? P->c1 + 9 * i : NULL,
                                     P->reg, R, t);
      c += r * r;
    }
  } else {
    for (int64_t i = 0; i < P->n; ++i) {
      const double r = pnec_oracle_residual(P->mode, P->b1 + 3 * i, P->b2 + 3 * i,
                                            P->c2 ? P->c2 + 9 * i : NULL,
                                            P->c1 ? P->c1 + 9 * i : NULL, P->reg, x->theta,
                                            x->phi, x->q);
      c += r * r;
    }
  }
  *cost = 0.5 * c;
  return isfinite(*cost);
}

/* cost, H = J'J (5x5 row-major, full), g = J'r in the tangent space */
static int eval_full(const problem_t *P, const state_t *x, double *cost, double H[25],
                     double g[5]) {
  double Pj[12];
  pose_uniforms U;
  quat_plus_jacobian(x->q, Pj);
  make_uniforms(x->theta, x->phi, x->q, &U);
  double c = 0.0;
  memset(H, 0, 25 * sizeof(double));
  memset(g, 0, 5 * sizeof(double));
  for (int64_t i = 0; i < P->n; ++i) {
    const double *f1 = P->b1 + 3 * i, *f2 = P->b2 + 3 * i;
    const double *c2 = P->c2 ? P->c2 + 9 * i : NULL, *c1 = P->c1 ? P->c1 + 9 * i : NULL;
    double r, J[5];
    if (P->jac_mode == PNEC_ORACLE_JAC_ANALYTIC)
      analytic_row(P->mode, f1, f2, c2, c1, P->reg, &U, &r, J);
    else
      numeric_row(P->mode, f1, f2, c2, c1, P->reg, x->theta, x->phi, x->q, Pj, &r, J);
    c += r * r;
    for (int a = 0; a < 5; ++a) {
      g[a] += J[a] * r;
      for (int b = 0; b < 5; ++b) H[5 * a + b] += J[a] * J[b];
    }
  }
  *cost = 0.5 * c;
  int ok = isfinite(*cost);
  for (int a = 0; a < 5; ++a) ok = ok && isfinite(g[a]);
  for (int a = 0; a < 25; ++a) ok = ok && isfinite(H[a]);
  return ok;
}

/* x_plus = Plus(x, delta): theta, phi Euclidean; quaternion per EigenQuaternionManifold::Plus */
static void state_plus(const state_t *x, const double d[5], state_t *y) {
  y->theta = x->theta + d[0];
  y->phi = x->phi + d[1];
  const double nd = sqrt(d[2] * d[2] + d[3] * d[3] + d[4] * d[4]);
  if (nd == 0.0) {
    memcpy(y->q, x->q, sizeof(y->q));
    return;
  }
  const double sbd = sin(nd) / nd;
  const double ax = sbd * d[2], ay = sbd * d[3], az = sbd * d[4], aw = cos(nd);
  const double bx = x->q[0], by = x->q[1], bz = x->q[2], bw = x->q[3];
  y->q[0] = aw * bx + ax * bw + ay * bz - az * by;
  y->q[1] = aw * by - ax * bz + ay * bw + az * bx;
  y->q[2] = aw * bz + ax * by - ay * bx + az * bw;
  y->q[3] = aw * bw - ax * bx - ay * by - az * bz;
}

/* 5x5 SPD solve A y = b by Cholesky; returns 0 if not positive definite / not finite */
static int chol_solve5(const double A[25], const double b[5], double y[5]) {
  double L[25];
  memset(L, 0, sizeof(L));
  for (int j = 0; j < 5; ++j) {
    double d = A[5 * j + j];
    for (int k = 0; k < j; ++k) d -= L[5 * j + k] * L[5 * j + k];
    if (!(d > 0.0) || !isfinite(d)) return 0;
    const double l = sqrt(d);
    L[5 * j + j] = l;
    for (int i = j + 1; i < 5; ++i) {
      double s = A[5 * i + j];
      for (int k = 0; k < j; ++k) s -= L[5 * i + k] * L[5 * j + k];
      L[5 * i + j] = s / l;
    }
  }
  double z[5];
  for (int i = 0; i < 5; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[5 * i + k] * z[k];
    z[i] = s / L[5 * i + i];
  }
  for (int i = 4; i >= 0; --i) {
    double s = z[i];
    for (int k = i + 1; k < 5; ++k) s -= L[5 * k + i] * y[k];
    y[i] = s / L[5 * i + i];
  }
  for (int i = 0; i < 5; ++i)
    if (!isfinite(y[i])) return 0;
  return 1;
}

/* Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy [EXT], default options. */
/* diagnostics (test tooling): how many LM steps reached the accept / reject decision, and how many were rejected */
static int g_lm_diagnostics = 0; /* pnec_oracle_lm_diagnostics(1) switches the counters below on */
void pnec_oracle_lm_diagnostics(int on) { g_lm_diagnostics = on; }
static long long g_lm_steps_evaluated = 0, g_lm_steps_rejected = 0, g_lm_invalid_steps = 0;
long long pnec_oracle_lm_invalid_steps(int reset) {
  const long long v = g_lm_invalid_steps;
  if (reset) g_lm_invalid_steps = 0;
  return v;
}
static long long g_lm_promise[2][24][2]; /* [after accepted-or-first | after rejected][decade of model_change / cost, -24 .. -1][accepted | rejected] */
void pnec_oracle_lm_promise_histogram(long long out[96]) { memcpy(out, g_lm_promise, sizeof(g_lm_promise)); memset(g_lm_promise, 0, sizeof(g_lm_promise)); }
static long long g_lm_transitions[3][2]; /* [first step | after an accepted step | after a rejected step][accepted | rejected] */
void pnec_oracle_lm_step_counts(int reset, long long out[8]) {
  out[0] = g_lm_steps_evaluated;
  out[1] = g_lm_steps_rejected;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 2; ++b) out[2 + 2 * a + b] = g_lm_transitions[a][b];
  if (reset) {
    g_lm_steps_evaluated = g_lm_steps_rejected = 0;
    memset(g_lm_transitions, 0, sizeof(g_lm_transitions));
  }
}

static int minimise(const problem_t *P, const pnec_oracle_options *o, state_t *x,
                    double *cost_out, int32_t *iters_out) {
  double cost, H[25], g[5], scale[5], Hs[25], gs[5], diag[5];
  int iteration = 0;
  *iters_out = 0;
  if (!eval_full(P, x, &cost, H, g)) {
    *cost_out = cost;
    return PNEC_ORACLE_TERM_BAD_INITIAL;
  }
  for (int j = 0; j < 5; ++j)
    scale[j] = o->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[6 * j])) : 1.0;

#define RESCALE()                                                        \
  do {                                                                   \
    gmax = 0.0;                                                          \
    for (int a = 0; a < 5; ++a) {                                        \
      gs[a] = g[a] * scale[a];                                           \
      if (fabs(g[a]) > gmax) gmax = fabs(g[a]);                          \
      for (int b = 0; b < 5; ++b) Hs[5 * a + b] = H[5 * a + b] * scale[a] * scale[b]; \
    }                                                                    \
  } while (0)

  double gmax;
  RESCALE();
  double x_norm = state_norm(x);
  double radius = o->initial_trust_region_radius, decrease_factor = 2.0;
  /* step_is_successful = 1 at iteration zero [EXT, recalled]: TrustRegionMinimizer::IterationZero() ends with
   * iteration_summary_.step_is_valid = iteration_summary_.step_is_successful = true, so the first
   * FinalizeIterationAndCheckIfMinimizerCanContinue() tests the gradient tolerance at the START point -- Ceres 1.x made
   * that test explicitly in front of its loop, and a solve started at a stationary point still reports
   * "Gradient tolerance reached" after iteration 0 alone. */
  int reuse_diagonal = 0, num_invalid = 0, step_is_successful = 1;
  int term;
  int prev_outcome = 0; /* diagnostics: 0 first step, 1 after an accepted, 2 after a rejected one */

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (iteration >= o->max_num_iterations) { term = PNEC_ORACLE_TERM_MAX_ITERATIONS; break; }
    if (o->check_convergence && step_is_successful && gmax <= o->gradient_tolerance) {
      term = PNEC_ORACLE_TERM_GRADIENT_TOL; break;
    }
    if (radius < o->min_trust_region_radius) { term = PNEC_ORACLE_TERM_MIN_RADIUS; break; }
    ++iteration;
    step_is_successful = 0;

    /* LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diagonal)
      for (int j = 0; j < 5; ++j)
        diag[j] = fmin(fmax(Hs[6 * j], o->min_lm_diagonal), o->max_lm_diagonal);
    double A[25], y[5], step[5];
    memcpy(A, Hs, sizeof(A));
    for (int j = 0; j < 5; ++j) A[6 * j] += diag[j] / radius;
    int valid = chol_solve5(A, gs, y);
    double model_cost_change = 0.0;
    if (valid) {
      for (int j = 0; j < 5; ++j) step[j] = -y[j];
      /* -(J s)'(r + J s / 2) = -(s'g + 1/2 s'Hs) */
      double sg = 0.0, sHs = 0.0;
      for (int a = 0; a < 5; ++a) {
        sg += step[a] * gs[a];
        for (int b = 0; b < 5; ++b) sHs += step[a] * Hs[5 * a + b] * step[b];
      }
      model_cost_change = -(sg + 0.5 * sHs);
      valid = model_cost_change > 0.0;
    }
    if (!valid) {
      if (++num_invalid >= o->max_num_consecutive_invalid_steps) {
        term = PNEC_ORACLE_TERM_INVALID_STEPS; break;
      }
      /* [EXT, recalled] TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid():
       * radius_ *= 0.5; reuse_diagonal_ = true; -- decrease_factor_ belongs to StepRejected() alone */
      radius *= 0.5;
      reuse_diagonal = 1;
      if (g_lm_diagnostics) {
#pragma omp atomic
        g_lm_invalid_steps += 1;
      }
      continue;
    }
    num_invalid = 0;

    double delta[5];
    for (int j = 0; j < 5; ++j) delta[j] = step[j] * scale[j];
    state_t xc;
    state_plus(x, delta, &xc);
    double cost_c;
    if (!eval_cost(P, &xc, &cost_c)) cost_c = DBL_MAX;

    if (o->check_convergence) {
      const double step_norm = state_dist(x, &xc);
      if (step_norm <= o->parameter_tolerance * (x_norm + o->parameter_tolerance)) {
        term = PNEC_ORACLE_TERM_PARAMETER_TOL; break;
      }
      if (fabs(cost - cost_c) <= o->function_tolerance * cost) {
        term = PNEC_ORACLE_TERM_FUNCTION_TOL; break;
      }
    }
    const double rho = (cost - cost_c) / model_cost_change;
    if (g_lm_diagnostics) {   /* test tooling, off unless switched on: the timed baseline runs without it */
#pragma omp atomic
      g_lm_steps_evaluated += 1;
      if (!(rho > o->min_relative_decrease)) {
#pragma omp atomic
        g_lm_steps_rejected += 1;
      }
      const int now = rho > o->min_relative_decrease ? 0 : 1;
#pragma omp atomic
      g_lm_transitions[prev_outcome][now] += 1;
      /* the model's promise relative to the cost, by decade, against what the step turned out to be */
      int dec = (int)floor(log10(fmax(model_cost_change / fmax(cost, 1e-300), 1e-30))) + 24;
      dec = dec < 0 ? 0 : (dec > 23 ? 23 : dec);
#pragma omp atomic
      g_lm_promise[prev_outcome == 2 ? 1 : 0][dec][now] += 1;
      prev_outcome = 1 + now;
    }
    if (rho > o->min_relative_decrease) {
      *x = xc;
      x_norm = state_norm(x);
      if (!eval_full(P, x, &cost, H, g)) { term = PNEC_ORACLE_TERM_BAD_INITIAL; break; }
      RESCALE();
      step_is_successful = 1;
      const double c3 = (2.0 * rho - 1.0) * (2.0 * rho - 1.0) * (2.0 * rho - 1.0);
      radius = radius / fmax(1.0 / 3.0, 1.0 - c3);
      radius = fmin(o->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
    }
  }
#undef RESCALE
  *cost_out = cost;
  *iters_out = iteration;
  return term;
}

int pnec_oracle_solve(int mode, int64_t n, const double *bvs1, const double *bvs2,
                      const double *covs2, const double *covs1, double reg,
                      const double init_q[4], const double init_t[3],
                      const pnec_oracle_options *opt, double out_q[4], double out_t[3],
                      double *out_theta_phi, double *out_cost, int32_t *out_iterations) {
  pnec_oracle_options defaults;
  if (!opt) {
    pnec_oracle_default_options(&defaults);
    opt = &defaults;
  }
  problem_t P = {mode, opt->jacobian_mode, n, bvs1, bvs2, covs2, covs1, reg};
  state_t x;
  /* PNECCeres::InitValues(orientation, translation) -- pnec_ceres.cc:182-186 */
  memcpy(x.q, init_q, sizeof(x.q));
  pnec_oracle_angles_from_vec(init_t, &x.theta, &x.phi);
  double cost = 0.0;
  int32_t iters = 0;
  const int term = minimise(&P, opt, &x, &cost, &iters);
  /* PNECCeres::Result() -- pnec_ceres.cc:201-207 */
  const double qn = sqrt(x.q[0] * x.q[0] + x.q[1] * x.q[1] + x.q[2] * x.q[2] + x.q[3] * x.q[3]);
  for (int k = 0; k < 4; ++k) out_q[k] = x.q[k] / qn;
  translation_from_angles(x.theta, x.phi, out_t);
  if (out_theta_phi) {
    out_theta_phi[0] = x.theta;
    out_theta_phi[1] = x.phi;
  }
  if (out_cost) *out_cost = cost;
  if (out_iterations) *out_iterations = iters;
  return term;
}

int pnec_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void pnec_oracle_solve_batch(int mode, int64_t n_pairs, const int64_t *offsets,
                             const double *bvs1, const double *bvs2, const double *covs2,
                             const double *covs1, double reg, const double *init_q,
                             const double *init_t, int32_t n_hyp, const double *hyp_t,
                             const pnec_oracle_options *opt, int num_threads, double *out_q,
                             double *out_t, double *out_cost, int32_t *out_iterations,
                             int32_t *out_status) {
  if (n_hyp < 1) n_hyp = 1;
  const int64_t n_solves = n_pairs * (int64_t)n_hyp;
#ifdef _OPENMP
  if (num_threads <= 0) num_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads)
#else
  (void)num_threads;
#endif
  for (int64_t s = 0; s < n_solves; ++s) {
    const int64_t p = s / n_hyp;
    const int64_t o = offsets[p], n = offsets[p + 1] - offsets[p];
    const double *t0 = hyp_t ? hyp_t + 3 * s : init_t + 3 * p;
    double cost = 0.0;
    int32_t it = 0;
    const int term = pnec_oracle_solve(mode, n, bvs1 + 3 * o, bvs2 + 3 * o,
                                       covs2 ? covs2 + 9 * o : NULL,
                                       covs1 ? covs1 + 9 * o : NULL, reg, init_q + 4 * p, t0, opt,
                                       out_q + 4 * s, out_t + 3 * s, NULL, &cost, &it);
    if (out_cost) out_cost[s] = cost;
    if (out_iterations) out_iterations[s] = it;
    if (out_status) out_status[s] = term;
  }
}
