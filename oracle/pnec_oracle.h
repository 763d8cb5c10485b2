/*
 * pnec_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the reference's PNEC least-squares hot path.
 * It exists to CHECK the HIP product path; nothing under pnec_amd/ may include,
 * link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * PARITY STATUS: "parity unpinned" for the optimiser trajectory.
 *   - The objective (residual + propagated-variance weight) IS pinned: it is
 *     checked against golden vectors produced by importing the reference's
 *     scripts/pnec/common.py (tests/golden/make_golden.py, tests/test_oracle_golden.py) --
 *     Target and NEC directly; Host and Symmetrical through the Python's target energy at the
 *     transformed inputs where it is their denominator (residual_forms_golden.npz, round 6).
 *   - The Gauss-Newton/LM arithmetic lives in Ceres Solver, which is neither
 *     vendored under /root/reference nor consistently pinned (Dockerfile:17 says
 *     1.13.0, src/optimization/pnec_ceres.cc:103 needs the Manifold API, >= 2.1),
 *     and the reference has no tests for it.  The LM below restates Ceres 2.1's
 *     published trust-region/Levenberg-Marquardt algorithm.
 *
 * Reference files followed (paths relative to /root/reference):
 *   include/optimization/pnec_residual.h:50-150   residual functors (Host/Target/Symmetrical)
 *   include/optimization/nec_residual.h:47-68     NEC residual
 *   src/optimization/pnec_ceres.cc:70-207         problem set-up, parameterisation, Result()
 *   src/optimization/nec_ceres.cc:73-139          NEC twin
 *   src/common/common.cc:96-116,210-259           SkewFromVector, AnglesFromVec, metrics
 *   src/rel_pose_estimation/pnec.cc:350-411       CeresSolver / CeresSolverFull / NECCeresSolver
 */
#ifndef PNEC_ORACLE_H_
#define PNEC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* residual families (value-compatible with include/pnec_hip.h; a test asserts it) */
enum {
  PNEC_ORACLE_MODE_NEC = 0,    /* nec_residual.h:51-63 */
  PNEC_ORACLE_MODE_TARGET = 1, /* pnec_residual.h:86-104 (default of PNECCeres::Optimize) */
  PNEC_ORACLE_MODE_HOST = 2,   /* pnec_residual.h:55-72 */
  PNEC_ORACLE_MODE_SYM = 3     /* pnec_residual.h:120-142 (what pypnec.pyceres runs) */
};

/* termination codes (value-compatible with include/pnec_hip.h) */
enum {
  PNEC_ORACLE_TERM_FUNCTION_TOL = 0,
  PNEC_ORACLE_TERM_PARAMETER_TOL = 1,
  PNEC_ORACLE_TERM_GRADIENT_TOL = 2,
  PNEC_ORACLE_TERM_MAX_ITERATIONS = 3,
  PNEC_ORACLE_TERM_MIN_RADIUS = 4,
  PNEC_ORACLE_TERM_INVALID_STEPS = 5,
  PNEC_ORACLE_TERM_BAD_INITIAL = 6
};

enum {
  PNEC_ORACLE_JAC_NUMERIC_CENTRAL = 0, /* what the reference does (pnec_ceres.cc:92-101) */
  PNEC_ORACLE_JAC_ANALYTIC = 1         /* same LM, closed-form Jacobian (what the HIP path does) */
};

/* Subset of ceres::Solver::Options that the default-constructed optimiser uses
 * (pnec_ceres.cc:47, pnec.cc:355).  Defaults = Ceres 2.x defaults [EXT]. */
typedef struct pnec_oracle_options {
  int32_t max_num_iterations;                /* 50 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;                    /* 1 */
  int32_t check_convergence;                 /* 1; 0 = run exactly max_num_iterations LM iterations */
  int32_t jacobian_mode;                     /* PNEC_ORACLE_JAC_* */
  int32_t reserved;
  double function_tolerance;                 /* 1e-6 */
  double gradient_tolerance;                 /* 1e-10 */
  double parameter_tolerance;                /* 1e-8 */
  double initial_trust_region_radius;        /* 1e4 */
  double max_trust_region_radius;            /* 1e16 */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3 */
  double min_lm_diagonal;                    /* 1e-6 */
  double max_lm_diagonal;                    /* 1e32 */
} pnec_oracle_options;

void pnec_oracle_default_options(pnec_oracle_options *opt);

/* --- small pieces (exported so the tests can pin each one) ------------------------------ */

/* common.cc:103-116 */
void pnec_oracle_angles_from_vec(const double v[3], double *theta, double *phi);
/* Eigen::Quaterniond(Matrix3d): rotation matrix (row-major 9) -> quaternion xyzw */
void pnec_oracle_quat_from_rot(const double R[9], double q[4]);
/* Eigen::Quaterniond::toRotationMatrix() -- NOT normalising; q = xyzw; R row-major */
void pnec_oracle_rot_from_quat(const double q[4], double R[9]);
/* pnec_ceres.cc:192-207: normalised q -> R, (theta,phi) -> unit t */
void pnec_oracle_result(const double q[4], double theta, double phi, double R[9], double t[3]);
/* common.cc:210-214, degrees */
double pnec_oracle_rotational_difference_deg(const double R1[9], const double R2[9]);
/* common.cc:216-235, degrees */
double pnec_oracle_translational_difference_deg(const double t1[3], const double t2[3],
                                                int both_directions);
/* common.cc:237-259: mean of n^2/(g' Sigma g), NO regularisation; covs AoS col-major 9 */
double pnec_oracle_cost_function(int64_t n, const double *bvs1, const double *bvs2,
                                 const double *covs, const double R[9], const double t[3]);

/* pnec::common::UnscentedTransform, src/common/common.cc:467-525 (camera_model: 0 = Omnidirectional,
 * 1 = Pinhole, the order of enum CameraModel at include/common/common.h:62).  mu 3, cov 9 and
 * K_inv 9 column-major, out 9 column-major. */
/* diagnostics: LM steps that reached Ceres' accept / reject decision since the last reset, how many were rejected, and
 * the outcome by predecessor: out[2 + 2 a + b], a = first step | after an accepted | after a rejected, b = accepted | rejected */
void pnec_oracle_lm_step_counts(int reset, long long out[8]);
/* ... and the steps that were INVALID (linear solve failed / model change <= 0: HandleInvalidStep) */
long long pnec_oracle_lm_invalid_steps(int reset);
/* Test tooling: multiply the central-difference step h = max(sqrt(eps), 1e-6 |x|) by `s` (default 1).  The derivative the
 * quotient approximates does not change (truncation error ~ h^2 f''' ~ 1e-16); its rounding error (~ eps |r| / h ~ 1e-8
 * relative) does.  Two runs that differ only in s show how far a solve's END POINT depends on that rounding -- i.e. on
 * the compiler, libm and FMA contraction of whoever builds the reference -- and not on the mathematics. */
void pnec_oracle_set_numeric_step_scale(double s);
/* ... counted only while switched on (off by default: the timed CPU baseline carries no test tooling) */
void pnec_oracle_lm_diagnostics(int on);
/* pnec::common::RotationBetweenPoints (common.cc:118-124) for unit vectors; out column-major */
void pnec_oracle_rotation_between_points(const double p1[3], const double p2[3], double out[9]);
void pnec_oracle_unscented_transform(const double mu[3], const double cov[9], const double K_inv[9],
                                     double kappa, int camera_model, double out[9]);
/* pnec::common::Unproject, common.cc:460-465: normalised K_inv (x, y, 1) */
void pnec_oracle_unproject(const double img_pt[2], const double K_inv[9], double out[3]);

/* One residual, literal form of the functors.  bv = 3 doubles; cov = 9 doubles (Eigen
 * column-major, as std::vector<Eigen::Matrix3d> stores them); cov1 only for SYM. */
double pnec_oracle_residual(int mode, const double f1[3], const double f2[3],
                            const double *cov2, const double *cov1, double reg, double theta,
                            double phi, const double q[4]);

/* Sum of squared residuals  sum_i r_i^2  (the energy of scripts/pnec/common.py:13-59). */
double pnec_oracle_energy(int mode, int64_t n, const double *bvs1, const double *bvs2,
                          const double *covs2, const double *covs1, double reg,
                          const double R[9], const double t[3]);

/* Residuals r[n], tangent-space Jacobian J[n*5] (row-major; columns theta, phi, delta_xyz of
 * EigenQuaternionManifold), cost = 1/2 sum r^2 at (theta, phi, q). */
void pnec_oracle_evaluate(int mode, int jacobian_mode, int64_t n, const double *bvs1,
                          const double *bvs2, const double *covs2, const double *covs1,
                          double reg, double theta, double phi, const double q[4], double *r,
                          double *J, double *cost);

/* --- the solver: PNECCeres::InitValues + Optimize + Result (pnec.cc:350-370) ------------ */
/* init_q xyzw (not nec. normalised, used as given), init_t any non-zero 3-vector.
 * out_q xyzw normalised; out_t unit; out_cost = 1/2 sum r^2 at the returned point.
 * Returns the termination code. */
int pnec_oracle_solve(int mode, int64_t n, const double *bvs1, const double *bvs2,
                      const double *covs2, const double *covs1, double reg,
                      const double init_q[4], const double init_t[3],
                      const pnec_oracle_options *opt, double out_q[4], double out_t[3],
                      double *out_theta_phi /* 2, may be NULL */, double *out_cost,
                      int32_t *out_iterations);

/* Batch driver (OpenMP over solves when compiled with -fopenmp; num_threads<=0 -> all).
 * offsets[B+1] into the concatenated AoS arrays; solve s = pair*n_hyp + h starts from
 * (init_q[pair], hyp_t[s]) when hyp_t != NULL, else (init_q[pair], init_t[pair]). */
void pnec_oracle_solve_batch(int mode, int64_t n_pairs, const int64_t *offsets,
                             const double *bvs1, const double *bvs2, const double *covs2,
                             const double *covs1, double reg, const double *init_q,
                             const double *init_t, int32_t n_hyp, const double *hyp_t,
                             const pnec_oracle_options *opt, int num_threads, double *out_q,
                             double *out_t, double *out_cost, int32_t *out_iterations,
                             int32_t *out_status);

int pnec_oracle_max_threads(void);

/* --- stages in front of the refinement (pnec_oracle_frontend.c) ----- */
/* symmetric 3x3 (row-major) -> eigenvalues ascending, eigenvectors in the columns of V (row-major);
 * each eigenvector's largest-magnitude component is positive */
void pnec_oracle_sym_eig3(const double A[9], double w[3], double V[9]);
/* common.cc:127-136 (skip_first = 1 reproduces the loop that starts at i = 1); M row-major */
void pnec_oracle_compose_m(int64_t n, const double *bvs1, const double *bvs2, const double R[9],
                           int skip_first, double M[9]);
/* common.cc:157-181 */
void pnec_oracle_translation_from_m(const double M[9], double t[3]);
/* common.cc:183-208; cov column-major */
double pnec_oracle_weight(const double f1[3], const double f2[3], const double t[3], const double R[9],
                          const double *cov, double reg, int host_frame);
void pnec_oracle_cayley_to_rot(const double v[3], double R[9]);
void pnec_oracle_rot_to_cayley(const double R[9], double v[3]);
/* Which iteration the eigenvalue minimisations run: 0 (default) damped Newton to ~1e-12 rad -- the device's default;
 * 1, 2 [EXT, from memory, unpinned]: the two recollections of opengv's own iteration (pnec_oracle_opengv.c: 1 = normalised
 * steepest descent with an adaptive step, stops ~1e-5 rad short; 2 = Eigen's Levenberg-Marquardt on the gradient of
 * lambda_min composed with the reduced Cayley rotation).  A process-wide switch for test tooling; set it before, not
 * during, a batch call. */
void pnec_oracle_set_eigensolver_scheme(int scheme);
/* RANSAC: hypothesis h + 1 starts from the last scored model's rotation (opengv's adapter side effect [EXT]); default off */
void pnec_oracle_set_ransac_chained_starts(int on);
int pnec_oracle_get_eigensolver_scheme(void);
/* scheme 2: Eigen's LevenbergMarquardtSpace::Status / nfev of the calling thread's last minimisation */
int pnec_oracle_es_last_info(void);
int pnec_oracle_es_last_nfev(void);
/* pnec_oracle_opengv.c: the 36 sums (opengv's xxF..zxF), lambda_min of M(v) composed from them (reduced != 0: with
 * math::cayley2rot_reduced) with its gradient / eigenvector / second eigenvalue (each optional), and the two minimisers
 * on the sums (v in/out; return = iterations) */
void pnec_oracle_sums36(int64_t n, const double *b1, const double *b2, double G[36]);
double pnec_oracle_es_value_grad_sums(const double G[36], const double v[3], int reduced, double *g, double *e_out,
                                      double *ev2);
int pnec_oracle_es_descent(const double G[36], double v[3], int *trips);
int pnec_oracle_es_lm(const double G[36], double v[3], int *nfev, int *info);
/* scheme 1 only, OFF by default: ge_main2's disturbed-restart loop around the descent (pnec_oracle_opengv.c: why off) */
void pnec_oracle_set_eigensolver_restart(int on);
int pnec_oracle_get_eigensolver_restart(void);
int pnec_oracle_es_descent_restarts(const double G[36], double v[3], uint64_t seed, uint64_t stream, int *trials);
/* opengv::relative_pose::eigensolver restated (Kneip-Lynen eigenvalue minimisation); R row-major */
int pnec_oracle_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                            double R_out[9], int32_t *iterations);
/* evaluations of the calling thread's last minimisation, counted as the device's quad spends them (one trip = a point
 * with its Hessian probes, or four step lengths): diagnostics for tools/sim_ransac_queue.py */
int pnec_oracle_es_last_trips(void);
/* scf.cc:53-72 (float division quirk), :43-51, :128-148 (with the alt_construct_E slip) */
void pnec_oracle_fibonacci_sphere(int samples, double *pts);
double pnec_oracle_obj_fun(const double t[3], int64_t n, const double *Ai, const double *Bi);
void pnec_oracle_scf(int64_t n, const double *Ai, const double *Bi, const double t0[3], int steps,
                     double t_out[3]);
/* pnec.cc:317-328; Ai, Bi row-major 3x3 per correspondence */
void pnec_oracle_build_ab(int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                          const double R[9], double reg, double *Ai, double *Bi);
/* pnec.cc:231-281 without RANSAC: rotation by the eigensolver, translation from ComposeM(i>=1) */
void pnec_oracle_nec_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                                 double R_out[9], double t_out[3]);
/* pnec.cc:239-272: RANSAC around the eigensolver (opengv restated; counter-based RNG) */
double pnec_oracle_rng_uniform(uint64_t seed, uint64_t pair, uint64_t hyp, uint64_t draw);
double pnec_oracle_reprojection_score(const double f1[3], const double f2[3], const double R[9],
                                      const double t[3]);
int pnec_oracle_ransac_eigensolver(int64_t n, const double *bvs1, const double *bvs2, const double R0[9],
                                   uint64_t seed, uint64_t pair_id, int max_iterations, int sample_size,
                                   double threshold, double R_out[9], double t_out[3], uint8_t *inlier_mask,
                                   int32_t *n_inliers, int32_t *iterations);
/* pnec.cc:283-348, literal: every round re-runs the eigensolver, every scf call runs its 10 steps */
void pnec_oracle_weighted_eigensolver_ex(int64_t n, const double *bvs1, const double *bvs2,
                                         const double *covs, const double R_init[9], const double t_init[3],
                                         double reg, int weighted_iterations,
                                         int device_early_exits /* 0 = the reference; 1 = the device's two
                                                                   declared early exits (NOT the reference) */,
                                         double R_out[9], double t_out[3]);
void pnec_oracle_weighted_eigensolver_batch(int64_t n_pairs, const int64_t *offsets, const double *bvs1,
                                            const double *bvs2, const double *covs, const double *R_init,
                                            const double *t_init, double reg, int weighted_iterations,
                                            int device_early_exits, int num_threads, double *R_out,
                                            double *t_out);
void pnec_oracle_weighted_eigensolver(int64_t n, const double *bvs1, const double *bvs2,
                                      const double *covs, const double R_init[9], const double t_init[3],
                                      double reg, int weighted_iterations, double R_out[9],
                                      double t_out[3]);

/* PNEC::Solve with the reference's default Options (pnec.cc:77-124: RANSAC eigensolver -> InlierExtraction ->
 * WeightedEigensolver -> CeresSolver) for a ragged batch, OpenMP over pairs; covs [M,9] column-major; pair p
 * draws as pair_id = first_pair_id + p; quaternions xyzw.  All outputs required. */
void pnec_oracle_solve_chain_batch(int64_t n_pairs, const int64_t *offsets, const double *bvs1,
                                   const double *bvs2, const double *covs, const double *init_q,
                                   uint64_t seed, uint64_t first_pair_id, int max_ransac_iterations,
                                   int sample_size, double threshold, double reg, int weighted_iterations,
                                   int num_threads, double *es_q, double *es_t, uint8_t *inlier_mask,
                                   int32_t *inlier_count, int32_t *ransac_iterations, double *w_q,
                                   double *w_t, double *out_q, double *out_t, int32_t *ls_iterations,
                                   int32_t *ls_status);

#ifdef __cplusplus
}
#endif
#endif /* PNEC_ORACLE_H_ */
