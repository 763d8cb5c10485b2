/*
 * pnec_hip.h -- C ABI of the MI355X-native PNEC pose solver (libpnec_hip.so).
 *
 * This is the drop-in boundary for the reference's rel_pose_estimation + optimization hot path.
 * The reference has no FFI layer; its seams are C++ classes and one pybind module.  Every entry
 * point below names the reference interface it replaces (paths relative to the reference repo):
 *
 *   reference seam                                              replaced by
 *   ----------------------------------------------------------  ---------------------------------
 *   PNECCeres::Optimize(bvs1,bvs2,covs,reg,frame)               pnec_hip_problem_create/upload +
 *     src/optimization/pnec_ceres.cc:70-111                       pnec_hip_solve (mode TARGET/HOST)
 *   PNECCeres::Optimize(bvs1,bvs2,covs1,covs2,reg)              ... mode SYM
 *     src/optimization/pnec_ceres.cc:113-168  (pypnec.pyceres, python/pypnec.cpp:50-66)
 *   NECCeres::Optimize(bvs1,bvs2)                               ... mode NEC
 *     src/optimization/nec_ceres.cc:73-101    (pypnec.pyceresnec, python/pypnec.cpp:68-82)
 *   PNECCeres::InitValues(q,t) / Result()                       init_q/init_t in, out_q/out_t out
 *     src/optimization/pnec_ceres.cc:182-186,201-207
 *   PNEC::CeresSolver / CeresSolverFull / NECCeresSolver        pnec_hip_solve over a batch of pairs
 *     src/rel_pose_estimation/pnec.cc:350-411
 *   ceres::Solver::Options (default-constructed, pnec_ceres.cc:47)   pnec_hip_options
 *   pnec::common::CostFunction  src/common/common.cc:237-259    pnec_hip_cost_function
 *   pnec::common::UnscentedTransform / Unproject  common.cc:460-525   pnec_hip_unscented_transform
 *   PNEC::Eigensolver (no RANSAC) / WeightedEigensolver  pnec.cc:231-348   pnec_hip_nec_eigensolver /
 *                                                               pnec_hip_weighted_eigensolver
 *
 * Conventions (same as the reference):
 *   - bearing vectors: 3 doubles each, unit norm; frame 1 = "host", frame 2 = "target".
 *   - covariances: 9 doubles each in Eigen column-major order (std::vector<Eigen::Matrix3d>);
 *     only the symmetric part matters.
 *   - quaternions: x,y,z,w (Eigen coeffs() order).  R takes frame-2 vectors into frame 1.
 *   - a batch holds n_pairs independent frame pairs; pair p owns correspondences
 *     [offsets[p], offsets[p+1]) of the concatenated arrays (ragged sizes allowed).
 *   - a "solve" is one (pair, hypothesis): s = pair * n_hyp + hypothesis.
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions return 0 on success or a
 * negative pnec_hip_status; pnec_hip_last_error() gives the message for the calling thread.
 * A handle (problem) is not thread-safe; distinct handles are independent.
 */
#ifndef PNEC_HIP_H_
#define PNEC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNEC_HIP_ABI_VERSION 7
#define PNEC_HIP_MAX_RANSAC_SAMPLE 16 /* largest Options::ransac_sample_size_ the RANSAC kernel is built for */

typedef enum pnec_hip_status {
  PNEC_HIP_OK = 0,
  PNEC_HIP_ERR_INVALID_ARGUMENT = -1,
  PNEC_HIP_ERR_HIP_RUNTIME = -2,   /* a hip* call failed (no device, OOM, launch failure ...) */
  PNEC_HIP_ERR_UNSUPPORTED = -3,
  PNEC_HIP_ERR_BUSY = -4           /* streaming handle: every slot holds a ticket that has not been collected */
} pnec_hip_status;

/* residual family == which reference functor the device evaluates */
typedef enum pnec_hip_mode {
  PNEC_HIP_MODE_NEC = 0,    /* include/optimization/nec_residual.h:51-63   */
  PNEC_HIP_MODE_TARGET = 1, /* include/optimization/pnec_residual.h:86-104 */
  PNEC_HIP_MODE_HOST = 2,   /* include/optimization/pnec_residual.h:55-72  */
  PNEC_HIP_MODE_SYM = 3     /* include/optimization/pnec_residual.h:120-142 */
} pnec_hip_mode;

/* per-solve termination code (out_status); what ceres::Solver::Summary would have said */
typedef enum pnec_hip_termination {
  PNEC_HIP_TERM_FUNCTION_TOL = 0,
  PNEC_HIP_TERM_PARAMETER_TOL = 1,
  PNEC_HIP_TERM_GRADIENT_TOL = 2,
  PNEC_HIP_TERM_MAX_ITERATIONS = 3,
  PNEC_HIP_TERM_MIN_RADIUS = 4,
  PNEC_HIP_TERM_INVALID_STEPS = 5,
  PNEC_HIP_TERM_BAD_INITIAL = 6 /* non-finite cost/Jacobian (e.g. NaN input); last iterate returned */
} pnec_hip_termination;

typedef enum pnec_hip_memspace {
  PNEC_HIP_MEM_HOST = 0,  /* pointer arguments are host memory; the call blocks until done */
  PNEC_HIP_MEM_DEVICE = 1 /* pointer arguments are device memory; the call is asynchronous on `stream` */
} pnec_hip_memspace;

/* The ceres::Solver::Options fields the reference's default-constructed optimiser relies on
 * (defaults = Ceres 2.x defaults), plus launch tuning.  Fill with pnec_hip_default_options(). */
typedef struct pnec_hip_options {
  int32_t max_num_iterations;                /* 50 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t jacobi_scaling;                    /* 1 */
  int32_t check_convergence;                 /* 1; 0 = exactly max_num_iterations LM iterations */
  int32_t corr_per_lane;                     /* 0 = auto; launch tuning: correspondences held per lane */
  int32_t waves_per_pair;                    /* 0 = auto; launch tuning: wavefronts cooperating on one solve */
  int32_t lds_corr_per_lane;                 /* launch tuning: how many of corr_per_lane live in LDS */
  int32_t flags;                             /* 0; a set of PNEC_HIP_OPT_* bits (below); an undefined bit:
                                                PNEC_HIP_ERR_INVALID_ARGUMENT.  (ABI <= 5 called this field `reserved`
                                                and defined bit 0 only.) */
  double function_tolerance;                 /* 1e-6 */
  double gradient_tolerance;                 /* 1e-10 */
  double parameter_tolerance;                /* 1e-8 */
  double initial_trust_region_radius;        /* 1e4 */
  double max_trust_region_radius;            /* 1e16 */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3 */
  double min_lm_diagonal;                    /* 1e-6 */
  double max_lm_diagonal;                    /* 1e32 */
} pnec_hip_options;

/* pnec_hip_options.flags */
#define PNEC_HIP_OPT_COUNT_PASSES 1 /* diagnostics: add the correspondence-passes this call executes -- in full /
                                        cost-only -- to pnec_hip_work_counters()[13] / [14] */
#define PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL 2
/* VERIFICATION mode (ABI 6): differentiate the way the reference does -- ceres::NumericDiffCostFunction<Functor, CENTRAL,
 * 1, 1, 1, 4> (src/optimization/pnec_ceres.cc:84-97; nec_ceres.cc:82-93): per AMBIENT parameter x_j of (theta, phi, qx, qy,
 * qz, qw) the residual at x_j +- h, h = max(sqrt(eps), 1e-6 |x_j|), J_j = (r+ - r-) / (2 h), the quaternion perturbed
 * component-wise WITHOUT renormalisation (toRotationMatrix of a non-unit quaternion), then the 1x4 block through
 * EigenQuaternionManifold::PlusJacobian (4x3) [EXT] -- instead of the closed-form Jacobian.  Thirteen residual evaluations
 * per correspondence and pass, streamed from HBM (no on-chip residency, one 8-wavefront block per solve): ~20x slower
 * than the production kernel, and not meant to be anything else.  It exists so that the device can follow the reference's
 * TRAJECTORY on solves that are not converged when they stop (fixed iteration counts from far-off starts), where the
 * rounding of the difference quotient -- not the derivative -- decides the last digits of every step.  pnec_hip_solve
 * only (the streaming handle returns PNEC_HIP_ERR_UNSUPPORTED). */

typedef struct pnec_hip_problem pnec_hip_problem; /* opaque: a batch of pairs resident in HBM */

/* Which iteration stands in for opengv::relative_pose::eigensolver's eigenvalue minimisation -- called at
 * src/rel_pose_estimation/pnec.cc:274 (plain), :239-258 (RANSAC hypotheses + optimizeModelCoefficients) and :315 (the
 * weighted stage's rounds).  opengv is NOT in the reference tree: all three are restatements, the last two from memory
 * of its source (oracle/pnec_oracle_opengv.c says what is remembered and how surely; INTEGRATION.md 6 which to pick).
 *   NEWTON  damped Newton on lambda_min(M(R(v))) to ~1e-12 rad (rounds 1-4's only form; the fastest).
 *   DESCENT [EXT] normalised steepest descent, step 0.01 doubled up to 0.08 in the first iteration, halved while the
 *           value does not improve, stop at step < 1e-5 or 50 iterations: ends ~1e-5 rad short of the minimiser.
 *   LM      [EXT] Eigen/MINPACK Levenberg-Marquardt (ftol 5e-5, xtol 10 eps, maxfev 100) on the gradient of lambda_min
 *           of M composed with opengv's REDUCED Cayley rotation (no 1 / (1 + |v|^2)), forward-difference Jacobian: the
 *           root of that gradient -- 1e-9 (KITTI-like motion) .. 1e-5 rad (|v| ~ 0.3) from NEWTON's minimiser.
 * Under DESCENT and LM every RANSAC hypothesis is scored (no iteration cap that voids a model), and the weighted
 * stage runs the eigensolver in every round under DESCENT (each call moves the rotation a little further). */
typedef enum pnec_hip_eigensolver_scheme {
  PNEC_HIP_ES_NEWTON = 0,
  PNEC_HIP_ES_DESCENT = 1,
  PNEC_HIP_ES_LM = 2
} pnec_hip_eigensolver_scheme;

int pnec_hip_abi_version(void);
const char *pnec_hip_last_error(void);
int pnec_hip_device_count(int *count);
void pnec_hip_default_options(pnec_hip_options *opt);

/* Allocate HBM for a batch.  offsets: HOST int64[n_pairs+1], non-decreasing, offsets[0]==0.
 * mode fixes which arrays the batch carries: NEC bvs only; TARGET/HOST bvs + one covariance
 * array; SYM bvs + both. */
int pnec_hip_problem_create(int device, int mode, int64_t n_pairs, const int64_t *offsets,
                            pnec_hip_problem **out);
int pnec_hip_problem_destroy(pnec_hip_problem *p);

/* A batch that is shaped again and again without allocating: room for up to max_pairs pairs holding up to
 * max_corr correspondences in total; created empty (0 pairs).  pnec_hip_problem_reshape gives it a shape --
 * offsets as in pnec_hip_problem_create -- on `stream` (the index arrays are uploaded asynchronously; work
 * queued earlier on that stream still sees the old shape); the planes then hold garbage until filled.  Shapes
 * beyond the capacity return PNEC_HIP_ERR_INVALID_ARGUMENT.  What the per-frame callers of the reference need
 * (one PNEC::Solve per frame pair, frame2frame.cc:122-141): see pnec_hip_frame_* below, which is built on it. */
int pnec_hip_problem_create_capacity(int device, int mode, int64_t max_pairs, int64_t max_corr,
                                     pnec_hip_problem **out);
int pnec_hip_problem_reshape(pnec_hip_problem *p, int64_t n_pairs, const int64_t *offsets, void *stream);

/* Fill pairs [first_pair, first_pair+n_pairs) from arrays in the REFERENCE layout (AoS: bvs
 * 3 doubles, covs 9 doubles column-major per correspondence), pointing at the first
 * correspondence of `first_pair`.  `covs` is the single array of Optimize(bvs1,bvs2,covs,..)
 * [frame 2 for TARGET, frame 1 for HOST] or covs_2 of the symmetric overload; `covs_host` is
 * covs_1 of the symmetric overload (NULL otherwise).  space = where those arrays live.
 * HOST space stages the arrays in HBM first (15 / 24 doubles per correspondence on top of the 12 / 18 of the planes): a
 * capacity-shaped batch keeps that staging between calls while it is <= 256 MB (per-frame handles: nothing allocated per
 * call); larger stagings are borrowed from the library's buffer cache for the call (pnec_hip_release_cache frees it). */
int pnec_hip_problem_fill(pnec_hip_problem *p, int64_t first_pair, int64_t n_pairs,
                          const double *bvs1, const double *bvs2, const double *covs,
                          const double *covs_host, int space, void *stream);

/* Fused keypoint ingest: KeyPoint::Unproject (src/frames/keypoints.cc:49-62) for the keypoints of both
 * frames -- bearing = normalised K^-1 (u, v, 1); bearing covariance = UnscentedTransform(mu, 2x2 image
 * covariance, K^-1, kappa, Pinhole) (src/common/common.cc:460-525) -- written straight into the batch's
 * SoA planes: 56 B per correspondence cross the bus instead of 120 B, and no AoS 3x3 covariance is ever
 * stored.  Same bits as pnec_hip_unscented_transform followed by pnec_hip_problem_fill.
 *   pts1, pts2 [M,2]   pixel coordinates (KeyPoint::point_) of frame 1 / frame 2
 *   cov2 [M,3]         frame-2 image covariance (xx, xy, yy) (KeyPoint::img_covariance_); NULL for NEC batches
 *   cov1 [M,3]         frame-1 image covariance, SYM batches only (else NULL)
 *   K_inv [9] column-major, kappa (1.0 in the reference), camera_model must be 1 (Pinhole)
 * M = correspondences of pairs [first_pair, first_pair + n_pairs), pointing at the first of them. */
int pnec_hip_problem_fill_keypoints(pnec_hip_problem *p, int64_t first_pair, int64_t n_pairs, const double *pts1,
                                    const double *pts2, const double *cov2, const double *cov1, const double *K_inv,
                                    double kappa, int camera_model, int space, void *stream);

/* The batch's SoA planes as stored in HBM (pair blocks of round_up(N_p, 64)-double planes; the layout in
 * the header of pnec_capi.hip / DESIGN.md): size in doubles, and a copy out (tests, debugging). */
int64_t pnec_hip_problem_payload_doubles(const pnec_hip_problem *p);
int pnec_hip_problem_export_payload(const pnec_hip_problem *p, double *out, int space, void *stream);

int64_t pnec_hip_problem_num_pairs(const pnec_hip_problem *p);
int64_t pnec_hip_problem_num_correspondences(const pnec_hip_problem *p);
int64_t pnec_hip_problem_max_correspondences(const pnec_hip_problem *p);
/* bytes of bearing/covariance payload the solver reads per pass over the batch (algorithmic) */
int64_t pnec_hip_problem_payload_bytes(const pnec_hip_problem *p);
/* The batch's correspondence offsets, HOST int64[n_pairs+1] (what problem_create was given; for a
 * batch made by pnec_hip_problem_select: the offsets of the kept correspondences). */
int pnec_hip_problem_offsets(const pnec_hip_problem *p, int64_t *out);
int pnec_hip_problem_mode(const pnec_hip_problem *p);
int pnec_hip_problem_device(const pnec_hip_problem *p);
/* The scheme the STAGE calls on this batch use (pnec_hip_nec_eigensolver, pnec_hip_ransac_eigensolver,
 * pnec_hip_weighted_eigensolver; default NEWTON).  pnec_hip_solve_pipeline takes its own from the options.  Under
 * DESCENT weighted_iterations is limited to 16 (a minimiser is kept per round; more: PNEC_HIP_ERR_UNSUPPORTED).  NEWTON
 * and LM run a further minimisation only while the previous one stopped at its evaluation cap (the weights never change
 * from round to round, pnec.cc:297-300), at most 15 of them per pair: a pair still at the cap then keeps that rotation
 * for the remaining rounds.  A batch handed out by pnec_hip_problem_select_view follows its source's scheme. */
int pnec_hip_problem_set_eigensolver_scheme(pnec_hip_problem *p, int32_t scheme);
int pnec_hip_problem_eigensolver_scheme(const pnec_hip_problem *p);

/* RANSAC variants (ABI 7): a set of bits for pnec_hip_pipeline_options.ransac_flags (the chain) and
 * pnec_hip_problem_set_ransac_flags (what pnec_hip_ransac_eigensolver on the batch runs with; default 0).
 *   PNEC_HIP_RANSAC_CHAINED_STARTS  [EXT, recalled: opengv is not in the reference tree]  opengv's
 *     EigensolverSacProblem::getSelectedDistancesToModel leaves the model it scores in the adapter
 *     (_adapter.setR12(model.rotation)), and computeModelCoefficients starts from _adapter.getR12() + jitter: hypothesis
 *     h + 1 starts from the rotation of the last model scored, not from the initial rotation the call site hands over once
 *     (src/rel_pose_estimation/pnec.cc:235-252).  A sequential dependence between hypotheses: with this bit a round is
 *     ONE hypothesis per pair (sixteen side by side without) and the stage costs 20-60x (600 pairs x 256, 25 % gross
 *     mismatches: 1.4 -> 30 ms under scheme 0, 2.2 -> 131 ms under scheme 2); the draws (sample, jitter) of hypothesis h
 *     are the same either way, the hypothesis counts grow (a contaminated sample's minimum is a poor start for the next
 *     one).  A FIDELITY switch, checked against the CPU checker's same switch (tests/test_opengv_schemes.py: masks and
 *     counts identical for 97-99 % of the pairs over chains of 100+ dependent hypotheses); off by default because the
 *     difference is inside the noise of opengv's rand(). */
#define PNEC_HIP_RANSAC_CHAINED_STARTS 1
int pnec_hip_problem_set_ransac_flags(pnec_hip_problem *p, int32_t flags);
int pnec_hip_problem_ransac_flags(const pnec_hip_problem *p);

/* Run InitValues + Optimize + Result for every solve of the batch, entirely on the device.
 *   init_q  [n_pairs,4] xyzw     starting orientation per pair (used as given)
 *   init_t  [n_pairs,3]          starting translation per pair (any non-zero vector; ignored if hyp_t)
 *   n_hyp, hyp_t [n_pairs*n_hyp,3]  optional multi-hypothesis starts sharing init_q (hyp_t NULL -> n_hyp=1)
 *   reg                         regularisation (Options::regularization_, 1e-13 in the reference)
 *   out_q [S,4] normalised, out_t [S,3] unit, out_cost [S] (= 1/2 sum r^2 at the result),
 *   out_iterations [S], out_status [S] (pnec_hip_termination); S = n_pairs*n_hyp; any out may be NULL
 *   space: where init and out arrays live (HOST: blocking; DEVICE: async on stream).
 * With n_hyp > 1 the hypotheses of a pair share its payload on chip: pairs of up to 512 correspondences run two
 * hypotheses per wavefront (both LM steps at once), larger ones one block per pair and group of 2 / 4 / 8 hypotheses (the
 * group's LM steps at the same time) -- the same bits as n_hyp separate calls, 1.2x .. 1.6x their rate.  */
int pnec_hip_solve(pnec_hip_problem *p, const double *init_q, const double *init_t, int32_t n_hyp,
                   const double *hyp_t, double reg, const pnec_hip_options *opt, double *out_q,
                   double *out_t, double *out_cost, int32_t *out_iterations, int32_t *out_status,
                   int space, void *stream);

/* For each pair keep the hypothesis with the lowest out_cost (ties: lowest index).
 * best_index [n_pairs] int32 receives the hypothesis index.  DEVICE or HOST pointers per `space`. */
int pnec_hip_select_best(int64_t n_pairs, int32_t n_hyp, const double *cost, int32_t *best_index,
                         int space, int device, void *stream);

/* pnec::common::CostFunction (src/common/common.cc:237-259) for every pair: mean over the
 * pair's correspondences of n^2 / (g' Sigma g), no regularisation; pose given as q (xyzw,
 * normalised inside) and t.  Only for TARGET-mode problems.  out [n_pairs]. */
int pnec_hip_cost_function(pnec_hip_problem *p, const double *q, const double *t, double *out,
                           int space, void *stream);

/* PNEC::Eigensolver with use_ransac_ = false (src/rel_pose_estimation/pnec.cc:273-278) for every
 * pair: rotation by opengv-style eigenvalue minimisation (Kneip-Lynen; opengv is not in the
 * reference tree, so the published algorithm is restated) started at init_q, translation by
 * TranslationFromM(ComposeM(...)) (src/common/common.cc:127-136,157-181, including ComposeM's
 * skipped first correspondence).  init_q [n_pairs,4] xyzw -> out_q [n_pairs,4], out_t [n_pairs,3]. */
int pnec_hip_nec_eigensolver(pnec_hip_problem *p, const double *init_q, double *out_q, double *out_t,
                             int space, void *stream);

/* PNEC::Eigensolver with use_ransac_ = true (src/rel_pose_estimation/pnec.cc:239-272): RANSAC over
 * eigensolver hypotheses from `sample_size` random correspondences (Options::ransac_sample_size_ = 10),
 * at most `max_iterations` (Options::max_ransac_iterations_ = 5000) with the adaptive bound for 99 %
 * confidence, inlier threshold on the midpoint-triangulation reprojection score (1e-6 in the
 * reference, pnec.cc:248), eigensolver re-run on the inliers, translation by
 * TranslationFromM(ComposeM(inliers)).  opengv::sac::Ransac is restated (opengv is not in the tree);
 * its rand() draws are replaced by a counter-based hash of (seed, pair, hypothesis, draw).
 * out_inlier_mask [sum N] (1 = inlier, in the caller's correspondence order), out_inlier_count
 * [n_pairs], out_ransac_iterations [n_pairs] may each be NULL.  Pairs with fewer than sample_size
 * correspondences fall back to the plain eigensolver with every correspondence an inlier.
 * sample_size > PNEC_HIP_MAX_RANSAC_SAMPLE (16) returns PNEC_HIP_ERR_UNSUPPORTED. */
int pnec_hip_ransac_eigensolver(pnec_hip_problem *p, const double *init_q, uint64_t seed,
                                int32_t max_iterations, int32_t sample_size, double threshold, double *out_q,
                                double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count,
                                int32_t *out_ransac_iterations, int space, void *stream);

/* PNEC::InlierExtraction (src/rel_pose_estimation/pnec.cc:210-229): a new batch holding, pair by pair
 * and in order, the correspondences whose mask byte is non-zero.  Done on the device: DEVICE-space calls
 * are asynchronous on `stream` and nothing is read back -- the new batch keeps the source's capacity and
 * its pair sizes stay in HBM until a host-side number is asked for (pnec_hip_problem_offsets /
 * _num_correspondences / _max_correspondences / _payload_bytes wait for the stream and fetch them).
 * HOST-space calls block (the caller may reuse `mask` on return). */
int pnec_hip_problem_select(pnec_hip_problem *src, const uint8_t *mask, int space, void *stream,
                            pnec_hip_problem **out);

/* The same InlierExtraction into the batch's CACHED target (the one pnec_hip_solve_pipeline compacts into): nothing is
 * allocated after the first call on `src`, nothing is to be destroyed.  *out is owned by `src` and valid until the next
 * select_view / solve_pipeline on `src`, a re-shape that outgrows it, or src's destruction.  For callers that run the
 * stages of PNEC::Solve one by one per frame (the timed overloads, pnec.cc:135-208) on a persistent batch. */
int pnec_hip_problem_select_view(pnec_hip_problem *src, const uint8_t *mask, int space, void *stream,
                                 pnec_hip_problem **out);

/* PNEC::WeightedEigensolver (src/rel_pose_estimation/pnec.cc:283-348) for every pair of a
 * TARGET-mode problem: (weighted_iterations - 1) rounds of { weights from the INITIAL pose x 1e-8,
 * eigensolver on the weighted bearings, 500-direction Fibonacci search of obj_fun
 * (src/optimization/scf.cc:43-72), 10 scf steps (scf.cc:128-148) }.  Options::weighted_iterations_
 * is 10 in the reference. */
int pnec_hip_weighted_eigensolver(pnec_hip_problem *p, const double *init_q, const double *init_t,
                                  double reg, int32_t weighted_iterations, double *out_q, double *out_t,
                                  int space, void *stream);

/* ---- the whole PNEC::Solve chain, device-resident ------------------------------------------------
 * pnec::rel_pose_estimation::Options as PNEC::Solve reads it (include/rel_pose_estimation/pnec_config.h:
 * 46-65; pnec.cc:87,96,97,105,109,116,239,246,249,300,327,367).  Fill with
 * pnec_hip_default_pipeline_options() (the reference's defaults). */
typedef struct pnec_hip_pipeline_options {
  int32_t use_ransac;            /* 1     Options::use_ransac_ */
  int32_t use_nec;               /* 0     Options::use_nec_ */
  int32_t use_ceres;             /* 1     Options::use_ceres_ */
  int32_t weighted_iterations;   /* 10    Options::weighted_iterations_ */
  int32_t max_ransac_iterations; /* 5000  Options::max_ransac_iterations_ */
  int32_t ransac_sample_size;    /* 10    Options::ransac_sample_size_ (<= PNEC_HIP_MAX_RANSAC_SAMPLE) */
  int64_t first_pair_id;         /* 0     RANSAC draws: pair p of the batch samples as pair first_pair_id + p, so a
                                          rank that solves pairs [a, b) of a larger set passes a and the results do
                                          not depend on how the set was sharded (>= 0; ABI 2 had reserved[2] here) */
  double regularization;         /* 1e-13 Options::regularization_ */
  double ransac_threshold;       /* 1e-6  pnec.cc:248 */
  uint64_t ransac_seed;          /* 1     counter-based draws (see pnec_hip_ransac_eigensolver) */
  pnec_hip_options solver;       /* the refinement's ceres::Solver::Options; PNEC::CeresSolver and
                                    NECCeresSolver default-construct theirs (pnec.cc:355,399) */
  int32_t eigensolver_scheme;    /* 0     pnec_hip_eigensolver_scheme: which iteration every eigenvalue minimisation of
                                          the chain runs (ABI 5) */
  int32_t ransac_flags;          /* 0     PNEC_HIP_RANSAC_* bits (ABI 7; `reserved` until ABI 6); an undefined bit:
                                          PNEC_HIP_ERR_INVALID_ARGUMENT */
} pnec_hip_pipeline_options;
void pnec_hip_default_pipeline_options(pnec_hip_pipeline_options *opt);

/* PNEC::Solve (src/rel_pose_estimation/pnec.cc:77-124) for every pair of a batch: Eigensolver (with
 * RANSAC when use_ransac) -> InlierExtraction -> NECCeresSolver (use_nec) or WeightedEigensolver +
 * CeresSolver, each stage one launch over the batch on `stream`, the stages handing their results to
 * each other in HBM (no host round trip, no allocation after the first call on a batch).
 *   init_q [n_pairs,4] xyzw, init_t [n_pairs,3]   initial_pose per pair
 *   out_q [n_pairs,4], out_t [n_pairs,3]          the pose Solve returns
 *   out_inlier_mask [sum N] / out_inlier_count [n_pairs]  the `inliers` of the four-argument overload
 *                                                 (all zero without RANSAC: inliers.clear()); may be NULL
 * TARGET-mode problems (use_nec also accepts NEC-mode ones).  space as in pnec_hip_solve. */
int pnec_hip_solve_pipeline(pnec_hip_problem *p, const double *init_q, const double *init_t,
                            const pnec_hip_pipeline_options *opt, double *out_q, double *out_t,
                            uint8_t *out_inlier_mask, int32_t *out_inlier_count, int space, void *stream);

/* Launch-order hint of the RANSAC stage (opt-in, scheduling only: results never depend on it).  A launch ends with
 * its slowest wavefronts: the pairs that need a second and third round of hypotheses (about a tenth at 10 % outliers)
 * run two to three times as long as the rest and end the launch late when they are dispatched late.  With the hint
 * enabled the batch remembers every pair's RANSAC hypothesis count of the last call that ran RANSAC on it
 * (pnec_hip_ransac_eigensolver, pnec_hip_solve_pipeline) and the next call on the same number of pairs dispatches
 * the pairs that went beyond one round first, each sharing its wavefront with one that did not.  Meaningful when
 * pair i of the next call is pair i of this one again (other start poses, other seeds, a benchmark loop) or its
 * successor in a stream of frames (sequence i's next frame pair: outlier ratios persist from frame to frame); a
 * stale hint costs nothing but the gain.  The reference has no counterpart (opengv's RANSAC runs one pair at a
 * time, pnec.cc:231-281). */
int pnec_hip_problem_launch_order_hint(pnec_hip_problem *p, int32_t enable);

/* ---- several GPUs of one node, one process ----------------------------------------------------------------
 * The reference fans out at process level (scripts/run_simulation.sh:52-67, scripts/parallel_kitti.sh:60-69: one
 * process per experiment / sequence).  Frame pairs are independent, so a batch shards with no data-path exchange:
 * pnec_hip_partition gives contiguous ranges of pairs balanced by correspondence count (bounds[n_parts + 1]; part r
 * owns pairs [bounds[r], bounds[r+1]) -- the rule the multi-process bench uses, pnec_amd/distributed.py::partition);
 * pnec_hip_solve_pipeline_multi runs PNEC::Solve (as pnec_hip_solve_pipeline) on one batch per entry of `devices`,
 * one host thread and one stream each, from HOST arrays in the reference layout (as pnec_hip_problem_fill: bvs 3
 * doubles, covs 9 doubles column-major per correspondence, NULL covs = NEC-only data for use_nec) and writes every
 * pair's result into the caller's arrays; it returns when all shards are done.  A device may be listed more than
 * once.  RANSAC draws belong to the GLOBAL pair index, so the results do not depend on the device list. */
int pnec_hip_partition(int64_t n_pairs, const int64_t *offsets, int32_t n_parts, int64_t *bounds);
int pnec_hip_solve_pipeline_multi(int32_t n_devices, const int32_t *devices, int64_t n_pairs, const int64_t *offsets,
                                  const double *bvs1, const double *bvs2, const double *covs, const double *init_q,
                                  const double *init_t, const pnec_hip_pipeline_options *opt, double *out_q,
                                  double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count);

/* The persistent form (ABI 5): a handle that keeps one capacity-shaped batch and one stream per listed device alive, for
 * callers that solve batch after batch -- the one-shot call above creates, fills and destroys a batch per device per
 * call.  What north_star shards is the refinement (PNECCeres::Optimize over independent frame pairs), so the handle has
 * it (pnec_hip_multi_solve = pnec_hip_solve per shard, multi-hypothesis starts included) next to the whole chain
 * (pnec_hip_multi_solve_pipeline = pnec_hip_solve_pipeline per shard).
 *   create  devices[n_devices] (a device may be listed more than once); mode as pnec_hip_problem_create; the handle holds
 *           up to max_pairs pairs / max_corr correspondences in total, no pair larger than max_pair_corr (each device's
 *           batch is sized for max_corr / n_devices + max_pair_corr: what a partition balanced by correspondence count
 *           can give it).
 *   fill    HOST arrays in the reference layout (as pnec_hip_problem_fill); partitions the pairs (pnec_hip_partition),
 *           re-shapes the device batches in place and uploads every shard from a host thread of its own.
 *   solve / solve_pipeline   HOST arrays; every shard on its device and stream, side by side; return when all are done.
 *           Results do not depend on the device list (RANSAC draws belong to the GLOBAL pair index).
 * After the first call of each kind nothing is allocated (pnec_hip_alloc_counters counts the library's hipMalloc calls).
 * Not thread-safe: one handle per calling thread. */
typedef struct pnec_hip_multi pnec_hip_multi;
int pnec_hip_multi_create(int32_t n_devices, const int32_t *devices, int mode, int64_t max_pairs, int64_t max_corr,
                          int64_t max_pair_corr, pnec_hip_multi **out);
int pnec_hip_multi_destroy(pnec_hip_multi *m);
int32_t pnec_hip_multi_num_devices(const pnec_hip_multi *m);
int pnec_hip_multi_bounds(const pnec_hip_multi *m, int64_t *bounds /* [n_devices + 1]: the current partition */);
int pnec_hip_multi_fill(pnec_hip_multi *m, int64_t n_pairs, const int64_t *offsets, const double *bvs1, const double *bvs2,
                        const double *covs, const double *covs_host);
int pnec_hip_multi_solve(pnec_hip_multi *m, const double *init_q, const double *init_t, int32_t n_hyp, const double *hyp_t,
                         double reg, const pnec_hip_options *opt, double *out_q, double *out_t, double *out_cost,
                         int32_t *out_iterations, int32_t *out_status);
int pnec_hip_multi_solve_pipeline(pnec_hip_multi *m, const double *init_q, const double *init_t,
                                  const pnec_hip_pipeline_options *opt, double *out_q, double *out_t,
                                  uint8_t *out_inlier_mask, int32_t *out_inlier_count);

/* ---- streaming: one frame pair (or a few) per call, as the reference's odometry calls the solver ----
 * (Frame2Frame::PNECAlign -> PNEC::Solve once per frame, src/rel_pose_estimation/frame2frame.cc:122-141;
 * PNECCeres::Optimize once per pybind call, python/pypnec.cpp:55-65.)  A handle owns `slots` staging
 * slots in pinned, device-mapped host memory and one HIP stream; nothing is allocated per call.
 *   submit  copies the caller's reference-layout arrays (as in pnec_hip_problem_fill; offsets[n_pairs+1]
 *           with offsets[0] == 0) and start poses into a free slot and launches ONE kernel that reads
 *           them over PCIe, runs InitValues + Optimize + Result on chip and writes the result records
 *           back into the slot; returns a ticket at once.  Up to `slots` tickets may be outstanding; with
 *           every slot holding an uncollected ticket submit returns PNEC_HIP_ERR_BUSY (collect the oldest
 *           with wait first: nothing is ever dropped).  Pairs beyond the register-resident geometries (> 4096 correspondences; > 2048
 *           for SYM) are staged through a batch owned by the handle instead.
 *   poll    done = 1 once the ticket's results are in host memory (never blocks).
 *   wait    blocks (polling a flag the kernel raises -- no stream synchronisation), copies the results
 *           out (any pointer may be NULL) and frees the slot.  Tickets must be collected with wait.
 * max_corr / max_pairs bound ONE submit.  stream: NULL = a stream of the handle's own.
 * A handle is not thread-safe; use one per thread. */
typedef struct pnec_hip_stream pnec_hip_stream;
int pnec_hip_stream_create(int device, int32_t max_corr, int32_t max_pairs, int32_t slots, void *stream,
                           pnec_hip_stream **out);
int pnec_hip_stream_destroy(pnec_hip_stream *s);
int pnec_hip_stream_submit(pnec_hip_stream *s, int mode, int64_t n_pairs, const int64_t *offsets,
                           const double *bvs1, const double *bvs2, const double *covs, const double *covs_host,
                           const double *init_q, const double *init_t, double reg, const pnec_hip_options *opt,
                           int64_t *ticket);
int pnec_hip_stream_poll(pnec_hip_stream *s, int64_t ticket, int32_t *done);
int pnec_hip_stream_wait(pnec_hip_stream *s, int64_t ticket, double *out_q, double *out_t, double *out_cost,
                         int32_t *out_iterations, int32_t *out_status);

/* ---- per-frame PNEC::Solve: the WHOLE chain for one frame pair per call, nothing allocated per call ----
 * (Frame2Frame::PNECAlign -> PNEC::Solve, src/rel_pose_estimation/frame2frame.cc:122-141 -> pnec.cc:77-124.)
 * A handle owns a pinned, device-mapped staging block, a capacity-shaped batch of one pair (TARGET family) with
 * its cached scratch / InlierExtraction target / side stream, and one HIP stream.
 *   solve   bvs1, bvs2 [n,3], covs [n,9] column-major (may be NULL with use_nec), start pose (q xyzw, t) in HOST
 *           memory -> pose out, inlier mask [n] and count (either may be NULL; zeros without RANSAC).  Runs
 *           pnec_hip_solve_pipeline on the handle's batch -- the same launches as the batch call, so the result
 *           is bit-identical to pnec_hip_solve_pipeline on a one-pair batch -- and returns when it is done.
 *   load    only the ingest (arrays -> SoA planes of the handle's batch, asynchronous on the handle's stream);
 *           *problem is the handle's batch, valid until the next load / solve / destroy, for callers that run the
 *           stages one by one (the timed PNEC::Solve overloads, pnec.cc:135-208) on pnec_hip_frame_stream().
 *           The caller's arrays are copied into the handle's pinned staging block before load returns (they may be
 *           reused at once); load first waits for everything queued earlier on the handle's stream, because the
 *           previous ingest reads that staging block when it executes -- two loads in a row are safe, not overlapped.
 * n <= max_corr (pnec_hip_frame_capacity).  Not thread-safe: one handle per thread. */
typedef struct pnec_hip_frame pnec_hip_frame;
int pnec_hip_frame_create(int device, int64_t max_corr, void *stream, pnec_hip_frame **out);
int pnec_hip_frame_destroy(pnec_hip_frame *f);
int64_t pnec_hip_frame_capacity(const pnec_hip_frame *f);
void *pnec_hip_frame_stream(const pnec_hip_frame *f);
int pnec_hip_frame_load(pnec_hip_frame *f, int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                        pnec_hip_problem **problem);
int pnec_hip_frame_solve(pnec_hip_frame *f, int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                         const double *init_q, const double *init_t, const pnec_hip_pipeline_options *opt,
                         double *out_q, double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count);

/* Input side of the path: pnec::common::UnscentedTransform (src/common/common.cc:467-525) and
 * pnec::common::Unproject (:460-465) for n keypoints at once -- what KeyPoint::Unproject
 * (src/frames/keypoints.cc:49-62) and the simulator's GetFeatures (src/simulation/sim_common.cc:72-107)
 * do per point.  mu [n,3] image points (x, y, 1) [or (x, y, f) with K_inv = I], covs [n,9]
 * column-major 3x3 whose top-left 2x2 is the image-plane covariance (omnidirectional: the
 * tangent-plane covariance rotated to the bearing), K_inv [9] column-major, kappa (1.0 in the
 * reference), camera_model 0 = Omnidirectional, 1 = Pinhole (enum CameraModel, common.h:62).
 * out_covs [n,9] bearing covariances; out_bvs [n,3] unit bearings or NULL. */
int pnec_hip_unscented_transform(int64_t n, const double *mu, const double *covs, const double *K_inv,
                                 double kappa, int camera_model, double *out_bvs, double *out_covs,
                                 int space, int device, void *stream);

/* Name and launch geometry the auto-tuner would pick for this problem (for logs / profiles). */
int pnec_hip_describe_launch(const pnec_hip_problem *p, const pnec_hip_options *opt,
                             int32_t *corr_per_lane, int32_t *waves_per_pair,
                             int32_t *lds_corr_per_lane, int32_t *threads_per_block,
                             int32_t *resident);

/* Device-side unit checks of the cross-lane reduction (DPP + v_permlane*_swap), the 5x5 Cholesky,
 * the reciprocal / reciprocal-square-root refinements, the lean trigonometry and the front stages'
 * smallest-eigenpair route (characteristic-polynomial start + Rayleigh-quotient iteration against the
 * Jacobi sweeps, on generic, nearly degenerate, rank-deficient and badly scaled matrices).  0 = all good. */
int pnec_hip_selftest(int device);

/* Work counters of the front stages, for roofline accounting: how many evaluations of the eigenvalue function, scored
 * tiles, table builds, ... the launches since the last reset held (the indices: enum kWk* in pnec_frontend.hip; the
 * numbers depend on the data, not on the timing).  Only a library built with -DPNEC_WORK_COUNT counts
 * (tools/count_chain_work.py builds and runs one); the production build compiles the counting out and reports
 * *compiled_in = 0 and zeros.  Entries [13] and [14] work in every build: the correspondence-passes the refinement
 * executed in full (residual, weight, Jacobian, normal equations) and cost-only (after a rejected step and at the iteration
 * cap) in the pnec_hip_solve calls made with PNEC_HIP_OPT_COUNT_PASSES in pnec_hip_options.flags.  Waits for the device. */
int pnec_hip_work_counters(int device, int reset, uint64_t *out16, int32_t *compiled_in);

/* The library keeps freed device buffers for reuse (batches are created and destroyed per frame or per
 * frame set in a pipeline; hipMalloc/hipFree cost tens of microseconds for small buffers and far more for
 * GB-sized ones).  Cap: environment
 * variable PNEC_HIP_CACHE_MB (default 16384, 0 disables).  This call returns the cached buffers
 * of `device` (-1: all devices) to the driver; returns the number of bytes released. */
int64_t pnec_hip_release_cache(int device);
/* out4: hipMalloc calls made by the library's allocator so far | requests served from its cache | blocks handed out and
 * not yet freed | bytes sitting in the cache (all devices; for "nothing is allocated after warm-up" tests) */
int pnec_hip_alloc_counters(uint64_t *out4);

#ifdef __cplusplus
}
#endif
#endif /* PNEC_HIP_H_ */
