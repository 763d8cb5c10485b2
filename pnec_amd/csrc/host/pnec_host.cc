// pnec_host.cc -- implementation of the host facade over the C ABI.  See pnec_host.h.
#include "pnec_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace pnec {
namespace {

void Check(int rc) {
  if (rc != 0) throw HipError(rc, std::string("libpnec_hip: ") + pnec_hip_last_error());
}

// RAII over pnec_hip_problem
struct Problem {
  pnec_hip_problem *p = nullptr;
  Problem(int device, int mode, const std::vector<int64_t> &offsets) {
    Check(pnec_hip_problem_create(device, mode, (int64_t)offsets.size() - 1, offsets.data(), &p));
  }
  ~Problem() { pnec_hip_problem_destroy(p); }
  Problem(const Problem &) = delete;
  Problem &operator=(const Problem &) = delete;
};

Vector3d TranslationFromAngles(double theta, double phi) {
  return Vector3d(std::sin(theta) * std::cos(phi), std::sin(theta) * std::sin(phi), std::cos(theta));
}

// The per-thread streaming handle of a device (pnec_hip_stream: pinned staging + one HIP stream, created
// once): what makes a single Optimize() call cost one memcpy + one kernel launch instead of a batch
// object's create / fill / solve / destroy.  Grown (re-created) when a pair exceeds its capacity.
struct ThreadStream {
  pnec_hip_stream *s = nullptr;
  int device = -1;
  int32_t max_corr = 0;
  ~ThreadStream() { pnec_hip_stream_destroy(s); }
  pnec_hip_stream *Get(int dev, int64_t n) {
    if (!s || dev != device || n > max_corr) {
      pnec_hip_stream_destroy(s);
      s = nullptr;
      int32_t cap = 4096;
      while (cap < n) cap *= 2;
      Check(pnec_hip_stream_create(dev, cap, 1, 4, nullptr, &s));
      device = dev;
      max_corr = cap;
    }
    return s;
  }
};
thread_local ThreadStream g_stream;

// One solve through the ABI: host arrays in the reference's layout in, pose + summary out.
void SolveOne(int mode, const optimization::SolverOptions &options, const std::vector<Vector3d> &b1,
              const std::vector<Vector3d> &b2, const std::vector<Matrix3d> *covs,
              const std::vector<Matrix3d> *covs_host, double reg, Quaterniond &q, double &theta,
              double &phi, optimization::Summary &summary) {
  if (b1.size() != b2.size()) throw std::invalid_argument("bvs_1 and bvs_2 differ in size");
  if (covs && covs->size() != b1.size()) throw std::invalid_argument("covs and bvs differ in size");
  if (covs_host && covs_host->size() != b1.size()) throw std::invalid_argument("covs_1 and bvs differ in size");
  const int64_t offsets[2] = {0, (int64_t)b1.size()};
  pnec_hip_stream *stream = g_stream.Get(options.device, offsets[1]);
  const Vector3d t0 = TranslationFromAngles(theta, phi);
  const pnec_hip_options o = options.ToHip();
  int64_t ticket = 0;
  Check(pnec_hip_stream_submit(stream, mode, 1, offsets, b1.empty() ? nullptr : b1[0].data(),
                               b2.empty() ? nullptr : b2[0].data(),
                               covs && !covs->empty() ? (*covs)[0].data() : nullptr,
                               covs_host && !covs_host->empty() ? (*covs_host)[0].data() : nullptr, q.coeffs(),
                               t0.data(), reg, &o, &ticket));
  double out_q[4], out_t[3], cost = 0.0;
  int32_t it = 0, st = 0;
  Check(pnec_hip_stream_wait(stream, ticket, out_q, out_t, &cost, &it, &st));
  q = Quaterniond(out_q[3], out_q[0], out_q[1], out_q[2]);
  common::AnglesFromVec(Vector3d(out_t[0], out_t[1], out_t[2]), theta, phi);
  summary.final_cost = cost;
  summary.iterations = it;
  summary.termination = st;
}

}  // namespace

// ------------------------------------------------------------------------------------ common
namespace common {

Matrix3d SkewFromVector(const Vector3d &v) {
  Matrix3d S;
  S(0, 1) = -v[2]; S(0, 2) = v[1];
  S(1, 0) = v[2];  S(1, 2) = -v[0];
  S(2, 0) = -v[1]; S(2, 1) = v[0];
  return S;
}

void AnglesFromVec(const Vector3d &vector, double &theta, double &phi) {
  const double n = vector.norm();
  if (n == 0.0) {
    theta = 0.0;
    phi = 0.0;
    return;
  }
  theta = std::acos(vector[2] / n);
  phi = (std::fabs(theta) < 1e-10) ? 0.0 : std::atan2(vector[1] / n, vector[0] / n);
}

double RotationalDifference(const Matrix3d &rotation_1, const Matrix3d &rotation_2) {
  const Quaterniond q = Quaterniond(rotation_1.transpose() * rotation_2).normalized();
  const double n = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z());
  const double theta = (q.w() < 0.0) ? 2.0 * std::atan2(-n, -q.w()) : 2.0 * std::atan2(n, q.w());
  return std::fabs(theta) * 180.0 / M_PI;
}

double TranslationalDifference(const Vector3d &translation_1, const Vector3d &translation_2,
                               bool both_directions) {
  const double n1 = translation_1.norm(), n2 = translation_2.norm();
  if (n1 < 1e-10) return 90.0;  // the reference tests translation_1 twice (common.cc:220)
  const double c = translation_1.dot(translation_2) / (n1 * n2);
  double error = std::acos(c);
  if (both_directions) error = std::min(error, std::acos(-c));
  return error * 180.0 / M_PI;
}

Matrix3d ComposeM(const bearingVectors_t &bvs_1, const bearingVectors_t &bvs_2, const Matrix3d &rotation) {
  Matrix3d M;
  for (size_t i = 1; i < bvs_1.size(); ++i) {  // sic: the reference's loop starts at 1
    const Vector3d n = bvs_1[i].cross(rotation * bvs_2[i]);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) M(r, c) += n[r] * n[c];
  }
  return M;
}

// unit eigenvector of the smallest eigenvalue (cyclic Jacobi); largest-magnitude component positive
Vector3d TranslationFromM(const Matrix3d &M) {
  double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A[r][c] = 0.5 * (M(r, c) + M(c, r));
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
    const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
    if (off <= 1e-34 * dg || off == 0.0) break;
    const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
    for (int k = 0; k < 3; ++k) {
      const int p = P[k], q = Q[k];
      if (A[p][q] == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
      const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
      for (int r = 0; r < 3; ++r) {
        const double arp = A[r][p], arq = A[r][q];
        A[r][p] = c * arp - s * arq;
        A[r][q] = s * arp + c * arq;
      }
      for (int r = 0; r < 3; ++r) {
        const double apr = A[p][r], aqr = A[q][r];
        A[p][r] = c * apr - s * aqr;
        A[q][r] = s * apr + c * aqr;
      }
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[r][p], vrq = V[r][q];
        V[r][p] = c * vrp - s * vrq;
        V[r][q] = s * vrp + c * vrq;
      }
    }
  }
  int j = 0;
  if (A[1][1] < A[j][j]) j = 1;
  if (A[2][2] < A[j][j]) j = 2;
  Vector3d v(V[0][j], V[1][j], V[2][j]);
  int big = 0;
  for (int r = 1; r < 3; ++r)
    if (std::fabs(v[r]) > std::fabs(v[big])) big = r;
  if (v[big] < 0.0) v = -v;
  return v.normalized();
}

double Weight(const Vector3d &f1, const Vector3d &f2, const Vector3d &translation, const Matrix3d &rotation,
              const Matrix3d &covariance, double regularization, bool host_frame) {
  const Vector3d v = host_frame ? translation.cross(rotation * f2) : rotation.transpose() * translation.cross(f1);
  const double q = v.dot(covariance * v);
  return host_frame ? 1.0 / q : 1.0 / (q + regularization);
}

Vector3d Unproject(const double img_pt[2], const Matrix3d &K_inv) {
  return (K_inv * Vector3d(img_pt[0], img_pt[1], 1.0)).normalized();
}

std::vector<Matrix3d> UnscentedTransform(const std::vector<Vector3d> &mus, const std::vector<Matrix3d> &covs,
                                         const Matrix3d &K_inv, double kappa, CameraModel camera_model) {
  if (mus.size() != covs.size()) return covs;  // the reference warns and returns the input (common.cc:532-537)
  std::vector<Matrix3d> out(mus.size());
  if (!mus.empty())
    Check(pnec_hip_unscented_transform((int64_t)mus.size(), mus[0].data(), covs[0].data(), K_inv.data(), kappa,
                                       camera_model == Pinhole ? 1 : 0, nullptr, out[0].data(),
                                       PNEC_HIP_MEM_HOST, optimization::SolverOptions().device, nullptr));
  return out;
}

Matrix3d UnscentedTransform(const Vector3d &mu, const Matrix3d &cov, const Matrix3d &K_inv, double kappa,
                            CameraModel camera_model) {
  return UnscentedTransform(std::vector<Vector3d>{mu}, std::vector<Matrix3d>{cov}, K_inv, kappa, camera_model)[0];
}

double CostFunction(const bearingVectors_t &bvs_1, const bearingVectors_t &bvs_2,
                    const std::vector<Matrix3d> &covs, const SE3d &camera_pose) {
  if (bvs_1.size() != bvs_2.size() || bvs_1.size() != covs.size())
    throw std::invalid_argument("bvs_1, bvs_2 and covs differ in size");
  // empty input: the reference divides 0.0 by size() == 0 (common.cc:258) -> NaN
  if (bvs_1.empty()) return std::nan("");
  const std::vector<int64_t> offsets = {0, (int64_t)bvs_1.size()};
  Problem prob(optimization::SolverOptions().device, PNEC_HIP_MODE_TARGET, offsets);
  Check(pnec_hip_problem_fill(prob.p, 0, 1, bvs_1[0].data(), bvs_2[0].data(), covs[0].data(), nullptr,
                              PNEC_HIP_MEM_HOST, nullptr));
  const Quaterniond q(camera_pose.rotationMatrix());
  double out = 0.0;
  Check(pnec_hip_cost_function(prob.p, q.coeffs(), camera_pose.translation().data(), &out,
                               PNEC_HIP_MEM_HOST, nullptr));
  return out;
}

std::ostream &operator<<(std::ostream &os, const FrameTiming &ft) {
  const long fields[] = {(long)ft.id_,    ft.frame_loading_, ft.feature_creation_,      ft.nec_es_,
                         ft.it_es_,       ft.avg_it_es_,     ft.ceres_,
                         (long)ft.OptimizationTime(),        (long)ft.TotalTime()};
  const char *sep = "";
  for (long v : fields) {
    os << sep << v;
    sep = " ";
  }
  return os;
}

std::string FrameTiming::TimingRowUs() const {
  char buf[192];
  std::snprintf(buf, sizeof(buf), "%d %.1f %.1f %.1f %.1f %.1f", id_, nec_es_us_, it_es_us_, avg_it_es_us_, ceres_us_,
                OptimizationTimeUs());
  return buf;
}

std::ostream &operator<<(std::ostream &os, const Timing &timing) {
  os << FrameTiming::TimingHeader() << std::endl;
  for (const FrameTiming &ft : timing.frame_timings_) os << ft << std::endl;
  return os;
}

bool Timing::Save(const std::string &path) const {
  std::ofstream out(path, std::ios_base::trunc);
  if (!out) return false;
  out << *this;
  return (bool)out;
}

}  // namespace common

// -------------------------------------------------------------------------------- optimization
namespace optimization {

pnec_hip_options SolverOptions::ToHip() const {
  pnec_hip_options o;
  pnec_hip_default_options(&o);
  o.max_num_iterations = max_num_iterations;
  o.max_num_consecutive_invalid_steps = max_num_consecutive_invalid_steps;
  o.jacobi_scaling = jacobi_scaling ? 1 : 0;
  o.function_tolerance = function_tolerance;
  o.gradient_tolerance = gradient_tolerance;
  o.parameter_tolerance = parameter_tolerance;
  o.initial_trust_region_radius = initial_trust_region_radius;
  o.max_trust_region_radius = max_trust_region_radius;
  o.min_trust_region_radius = min_trust_region_radius;
  o.min_relative_decrease = min_relative_decrease;
  o.min_lm_diagonal = min_lm_diagonal;
  o.max_lm_diagonal = max_lm_diagonal;
  return o;
}

PNECCeres::PNECCeres() : orientation_(1.0, 0.0, 0.0, 0.0), theta_(0.0), phi_(0.0) {}
PNECCeres::PNECCeres(const SE3d &init, const SolverOptions &options) : options_(options) {
  orientation_ = init.unit_quaternion();
  common::AnglesFromVec(init.translation(), theta_, phi_);
}
// (the reference drops `options` in this overload, pnec_ceres.cc:57-59; kept here)
PNECCeres::PNECCeres(const Quaterniond &orientation, double theta, double phi,
                     const SolverOptions &options)
    : orientation_(orientation), theta_(theta), phi_(phi), options_(options) {}
PNECCeres::PNECCeres(const Quaterniond &orientation, const Vector3d &translation,
                     const SolverOptions &options)
    : orientation_(orientation), options_(options) {
  common::AnglesFromVec(translation, theta_, phi_);
}
PNECCeres::~PNECCeres() {}

void PNECCeres::Run(int mode, const std::vector<Vector3d> &b1, const std::vector<Vector3d> &b2,
                    const std::vector<Matrix3d> *covs, const std::vector<Matrix3d> *covs_host,
                    double reg) {
  SolveOne(mode, options_, b1, b2, covs, covs_host, reg, orientation_, theta_, phi_, summary_);
}

void PNECCeres::Optimize(const std::vector<Vector3d> &bvs_1, const std::vector<Vector3d> &bvs_2,
                         const std::vector<Matrix3d> &covs, double regularization,
                         common::NoiseFrame noise_frame) {
  Run(noise_frame == common::Host ? PNEC_HIP_MODE_HOST : PNEC_HIP_MODE_TARGET, bvs_1, bvs_2, &covs,
      nullptr, regularization);
}

void PNECCeres::Optimize(const std::vector<Vector3d> &bvs_1, const std::vector<Vector3d> &bvs_2,
                         const std::vector<Matrix3d> &covs_1, const std::vector<Matrix3d> &covs_2,
                         double regularization) {
  // PNECSymmetrical(bv_1, bv_2, cov_1, cov_2): cov_2 rides with R Sigma R', cov_1 with [R f2]x
  Run(PNEC_HIP_MODE_SYM, bvs_1, bvs_2, &covs_2, &covs_1, regularization);
}

void PNECCeres::InitValues(const Quaterniond orientation, double theta, double phi) {
  orientation_ = orientation;
  theta_ = theta;
  phi_ = phi;
}
void PNECCeres::InitValues(const SE3d &init) {
  orientation_ = init.unit_quaternion();
  common::AnglesFromVec(init.translation(), theta_, phi_);
}
void PNECCeres::InitValues(const Quaterniond &orientation, const Vector3d &translation) {
  orientation_ = orientation;
  common::AnglesFromVec(translation, theta_, phi_);
}
void PNECCeres::SetOptions(const SolverOptions &options) { options_ = options; }
Matrix3d PNECCeres::Orientation() const { return orientation_.normalized().toRotationMatrix(); }
Vector3d PNECCeres::Translation() const { return TranslationFromAngles(theta_, phi_); }
SE3d PNECCeres::Result() const { return SE3d(Orientation(), Translation()); }

NECCeres::NECCeres() : orientation_(1.0, 0.0, 0.0, 0.0), theta_(0.0), phi_(0.0) {}
NECCeres::NECCeres(const SE3d &init, const SolverOptions &options) : options_(options) {
  orientation_ = init.unit_quaternion();
  common::AnglesFromVec(init.translation(), theta_, phi_);
}
NECCeres::NECCeres(const Quaterniond &orientation, double theta, double phi, const SolverOptions &options)
    : orientation_(orientation), theta_(theta), phi_(phi), options_(options) {}
NECCeres::NECCeres(const Quaterniond &orientation, const Vector3d &translation,
                   const SolverOptions &options)
    : orientation_(orientation), options_(options) {
  common::AnglesFromVec(translation, theta_, phi_);
}
NECCeres::~NECCeres() {}
void NECCeres::Optimize(const std::vector<Vector3d> &bvs_1, const std::vector<Vector3d> &bvs_2) {
  SolveOne(PNEC_HIP_MODE_NEC, options_, bvs_1, bvs_2, nullptr, nullptr, 0.0, orientation_, theta_,
           phi_, summary_);
}
void NECCeres::InitValues(const Quaterniond orientation, double theta, double phi) {
  orientation_ = orientation;
  theta_ = theta;
  phi_ = phi;
}
void NECCeres::InitValues(const SE3d &init) {
  orientation_ = init.unit_quaternion();
  common::AnglesFromVec(init.translation(), theta_, phi_);
}
void NECCeres::InitValues(const Quaterniond &orientation, const Vector3d &translation) {
  orientation_ = orientation;
  common::AnglesFromVec(translation, theta_, phi_);
}
void NECCeres::SetOptions(const SolverOptions &options) { options_ = options; }
Matrix3d NECCeres::Orientation() const { return orientation_.normalized().toRotationMatrix(); }
Vector3d NECCeres::Translation() const { return TranslationFromAngles(theta_, phi_); }
SE3d NECCeres::Result() const { return SE3d(Orientation(), Translation()); }

}  // namespace optimization

// -------------------------------------------------------------------------- rel_pose_estimation
namespace rel_pose_estimation {

PNEC::PNEC(const Options &options) : options_(options) {}

SE3d PNEC::Solve(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                 const std::vector<Matrix3d> &projected_covs, const SE3d &initial_pose) {
  std::vector<int> inliers;
  return Solve(bvs1, bvs2, projected_covs, initial_pose, inliers);
}

namespace {
// The per-thread frame handle of a device (pnec_hip_frame: pinned staging, a capacity-shaped one-pair batch with
// its cached scratch, one HIP stream -- all created once): PNEC::Solve is called once per frame pair by the
// odometry (frame2frame.cc:122-141), and a batch object per call cost more than the chain's kernels take.
// Grown (re-created) when a pair exceeds its capacity.
struct ThreadFrame {
  pnec_hip_frame *f = nullptr;
  int device = -1;
  ~ThreadFrame() { pnec_hip_frame_destroy(f); }
  pnec_hip_frame *Get(int dev, int64_t n) {
    if (!f || dev != device || n > pnec_hip_frame_capacity(f)) {
      pnec_hip_frame_destroy(f);
      f = nullptr;
      int64_t cap = 2048;
      while (cap < n) cap *= 2;
      Check(pnec_hip_frame_create(dev, cap, nullptr, &f));
      device = dev;
    }
    return f;
  }
};
thread_local ThreadFrame g_frame;

// one pair resident on the device (the thread's frame handle's batch), shared by the stages of Solve
struct PairOnDevice {
  struct {
    pnec_hip_problem *p = nullptr;
  } prob;
  pnec_hip_frame *frame = nullptr;
  PairOnDevice(int device, const bearingVectors_t &b1, const bearingVectors_t &b2,
               const std::vector<Matrix3d> &covs) {
    if (b1.size() != b2.size() || b1.size() != covs.size())
      throw std::invalid_argument("bvs1, bvs2 and projected_covs differ in size");
    frame = g_frame.Get(device, (int64_t)b1.size());
    Check(pnec_hip_frame_load(frame, (int64_t)b1.size(), b1.empty() ? nullptr : b1[0].data(),
                              b2.empty() ? nullptr : b2[0].data(), covs.empty() ? nullptr : covs[0].data(), &prob.p));
  }
};

// the Options fields PNEC::Solve reads (pnec.cc:87,96,97,105,109,116,239,246,249,300,327,367); the
// refinement uses default solver options whatever Options::ceres_options_ says (quirk C1, pnec.cc:355)
pnec_hip_pipeline_options ToPipeline(const Options &o) {
  pnec_hip_pipeline_options p;
  pnec_hip_default_pipeline_options(&p);
  p.use_ransac = o.use_ransac_ ? 1 : 0;
  p.use_nec = o.use_nec_ ? 1 : 0;
  p.use_ceres = o.use_ceres_ ? 1 : 0;
  p.weighted_iterations = (int32_t)o.weighted_iterations_;
  p.max_ransac_iterations = o.max_ransac_iterations_;
  p.ransac_sample_size = o.ransac_sample_size_;
  p.regularization = o.regularization_;
  p.solver = optimization::SolverOptions().ToHip();
  p.eigensolver_scheme = o.eigensolver_scheme_;
  p.ransac_flags = o.ransac_chained_starts_ ? PNEC_HIP_RANSAC_CHAINED_STARTS : 0;
  return p;
}

SE3d PoseFromQT(const double q[4], const double t[3]) {
  return SE3d(Quaterniond(q[3], q[0], q[1], q[2]).toRotationMatrix(), Vector3d(t[0], t[1], t[2]));
}
}  // namespace

SE3d PNEC::Solve(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                 const std::vector<Matrix3d> &projected_covs, const SE3d &initial_pose,
                 std::vector<int> &inliers) {
  return SolveImpl(bvs1, bvs2, projected_covs, initial_pose, inliers, nullptr);
}
SE3d PNEC::Solve(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                 const std::vector<Matrix3d> &projected_covs, const SE3d &initial_pose,
                 common::FrameTiming &timing) {
  std::vector<int> inliers;
  return SolveImpl(bvs1, bvs2, projected_covs, initial_pose, inliers, &timing);
}
SE3d PNEC::Solve(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                 const std::vector<Matrix3d> &projected_covs, const SE3d &initial_pose,
                 std::vector<int> &inliers, common::FrameTiming &timing) {
  return SolveImpl(bvs1, bvs2, projected_covs, initial_pose, inliers, &timing);
}

SE3d PNEC::SolveImpl(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                     const std::vector<Matrix3d> &projected_covs, const SE3d &initial_pose,
                     std::vector<int> &inliers, common::FrameTiming *timing) {
  // pnec.cc:77-124 (timed twin :135-208), stage by stage on the device.
  using clock = std::chrono::high_resolution_clock;
  // (one reading per stage: the millisecond field is the microsecond reading truncated, as duration_cast does)
  auto us_since = [](clock::time_point t0) { return std::chrono::duration<double, std::micro>(clock::now() - t0).count(); };
  auto tic = clock::now();
  const Quaterniond q0(initial_pose.rotationMatrix());
  double q[4], t[3];
  if (!timing) {
    // the untimed overloads: the whole chain in one call on the thread's frame handle -- arrays in through
    // pinned staging, stages chained on the device, pose and inlier mask written straight back
    if (bvs1.size() != bvs2.size() || bvs1.size() != projected_covs.size())
      throw std::invalid_argument("bvs1, bvs2 and projected_covs differ in size");
    const pnec_hip_pipeline_options po = ToPipeline(options_);
    std::vector<uint8_t> mask(bvs1.size() ? bvs1.size() : 1, 0);
    pnec_hip_frame *frame = g_frame.Get(optimization::SolverOptions().device, (int64_t)bvs1.size());
    Check(pnec_hip_frame_solve(frame, (int64_t)bvs1.size(), bvs1.empty() ? nullptr : bvs1[0].data(),
                               bvs2.empty() ? nullptr : bvs2[0].data(),
                               projected_covs.empty() ? nullptr : projected_covs[0].data(), q0.coeffs(),
                               initial_pose.translation().data(), &po, q, t, mask.data(), nullptr));
    inliers.clear();
    if (options_.use_ransac_)
      for (size_t i = 0; i < bvs1.size(); ++i)
        if (mask[i]) inliers.push_back((int)i);
    return PoseFromQT(q, t);
  }
  // the timed overloads run stage by stage (each stage's wall time is what FrameTiming reports)
  PairOnDevice dev(optimization::SolverOptions().device, bvs1, bvs2, projected_covs);
  void *const st = pnec_hip_frame_stream(dev.frame);   // the ingest is queued there: the stages follow it
  pnec_hip_problem *stage = dev.prob.p;   // the batch the later stages run on (inliers under RANSAC)
  pnec_hip_problem *selected = nullptr;   // the handle's cached InlierExtraction target: nothing to destroy
  Check(pnec_hip_problem_set_eigensolver_scheme(dev.prob.p, options_.eigensolver_scheme_));
  Check(pnec_hip_problem_set_ransac_flags(dev.prob.p, options_.ransac_chained_starts_ ? PNEC_HIP_RANSAC_CHAINED_STARTS : 0));
  // ES_solution = Eigensolver(bvs1, bvs2, initial_pose, inliers)
  inliers.clear();
  if (options_.use_ransac_) {
    std::vector<uint8_t> mask(bvs1.size() ? bvs1.size() : 1);
    Check(pnec_hip_ransac_eigensolver(dev.prob.p, q0.coeffs(), /*seed*/ 1, options_.max_ransac_iterations_,
                                      options_.ransac_sample_size_, /*threshold pnec.cc:248*/ 1.0e-6, q, t,
                                      mask.data(), nullptr, nullptr, PNEC_HIP_MEM_HOST, st));
    for (size_t i = 0; i < bvs1.size(); ++i)
      if (mask[i]) inliers.push_back((int)i);
    // InlierExtraction (pnec.cc:210-229)
    Check(pnec_hip_problem_select_view(dev.prob.p, mask.data(), PNEC_HIP_MEM_HOST, st, &selected));
    Check(pnec_hip_problem_set_eigensolver_scheme(selected, options_.eigensolver_scheme_));
    stage = selected;
  } else {
    Check(pnec_hip_nec_eigensolver(dev.prob.p, q0.coeffs(), q, t, PNEC_HIP_MEM_HOST, st));
  }
  if (timing) { timing->nec_es_us_ = us_since(tic); timing->nec_es_ = (long)(timing->nec_es_us_ / 1000.0); }
  const pnec_hip_options o = optimization::SolverOptions().ToHip();
  double oq[4], ot[3];
  if (options_.use_nec_) {
    if (timing) timing->ceres_ = 0;
    if (!options_.use_ceres_) return PoseFromQT(q, t);
    tic = clock::now();
    // NECCeresSolver on the (inlier) bearings: the NEC residual ignores the covariance planes, but
    // the kernel family is fixed by the batch, so the inlier bearings go into a NEC batch
    std::vector<Vector3d> b1, b2;
    if (options_.use_ransac_) {
      for (int i : inliers) { b1.push_back(bvs1[(size_t)i]); b2.push_back(bvs2[(size_t)i]); }
    } else {
      b1 = bvs1; b2 = bvs2;
    }
    const SE3d nec = NECCeresSolver(b1, b2, PoseFromQT(q, t));
    if (timing) { timing->ceres_us_ = us_since(tic); timing->ceres_ = (long)(timing->ceres_us_ / 1000.0); }
    return nec;
  }
  double qi[4], ti[3];
  // it_es_ is written in every branch, avg_it_es_ only when the weighted stage runs (pnec.cc:178-195)
  if (timing) timing->it_es_ = 0;
  if (options_.weighted_iterations_ > 1) {
    tic = clock::now();
    Check(pnec_hip_weighted_eigensolver(stage, q, t, options_.regularization_,
                                        (int32_t)options_.weighted_iterations_, qi, ti, PNEC_HIP_MEM_HOST,
                                        st));
    if (timing) {
      timing->it_es_us_ = us_since(tic);
      timing->it_es_ = (long)(timing->it_es_us_ / 1000.0);
      timing->avg_it_es_ = timing->it_es_ / (long)options_.weighted_iterations_;
      timing->avg_it_es_us_ = timing->it_es_us_ / (double)options_.weighted_iterations_;
    }
  } else if (options_.weighted_iterations_ == 1) {
    std::memcpy(qi, q, sizeof(qi));
    std::memcpy(ti, t, sizeof(ti));
  } else {
    std::memcpy(qi, q0.coeffs(), sizeof(qi));
    std::memcpy(ti, initial_pose.translation().data(), sizeof(ti));
  }
  if (timing) timing->ceres_ = 0;
  if (!options_.use_ceres_) return PoseFromQT(qi, ti);
  // CeresSolver: default-constructed optimiser, Target frame (pnec.cc:355,366)
  tic = clock::now();
  Check(pnec_hip_solve(stage, qi, ti, 1, nullptr, options_.regularization_, &o, oq, ot, nullptr, nullptr,
                       nullptr, PNEC_HIP_MEM_HOST, st));
  if (timing) { timing->ceres_us_ = us_since(tic); timing->ceres_ = (long)(timing->ceres_us_ / 1000.0); }
  return PoseFromQT(oq, ot);
}

SE3d PNEC::Eigensolver(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                       const SE3d &initial_pose, std::vector<int> &inliers) {
  inliers.clear();
  const std::vector<int64_t> offsets = {0, (int64_t)bvs1.size()};
  Problem prob(optimization::SolverOptions().device, PNEC_HIP_MODE_NEC, offsets);
  if (!bvs1.empty())
    Check(pnec_hip_problem_fill(prob.p, 0, 1, bvs1[0].data(), bvs2[0].data(), nullptr, nullptr,
                                PNEC_HIP_MEM_HOST, nullptr));
  Check(pnec_hip_problem_set_eigensolver_scheme(prob.p, options_.eigensolver_scheme_));
  Check(pnec_hip_problem_set_ransac_flags(prob.p, options_.ransac_chained_starts_ ? PNEC_HIP_RANSAC_CHAINED_STARTS : 0));
  const Quaterniond q0(initial_pose.rotationMatrix());
  double q[4], t[3];
  if (options_.use_ransac_) {
    std::vector<uint8_t> mask(bvs1.size() ? bvs1.size() : 1);
    Check(pnec_hip_ransac_eigensolver(prob.p, q0.coeffs(), 1, options_.max_ransac_iterations_,
                                      options_.ransac_sample_size_, 1.0e-6, q, t, mask.data(), nullptr, nullptr,
                                      PNEC_HIP_MEM_HOST, nullptr));
    for (size_t i = 0; i < bvs1.size(); ++i)
      if (mask[i]) inliers.push_back((int)i);
  } else {
    Check(pnec_hip_nec_eigensolver(prob.p, q0.coeffs(), q, t, PNEC_HIP_MEM_HOST, nullptr));
  }
  return PoseFromQT(q, t);
}

SE3d PNEC::WeightedEigensolver(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                               const std::vector<Matrix3d> &projected_covariances,
                               const SE3d &initial_pose) {
  PairOnDevice dev(optimization::SolverOptions().device, bvs1, bvs2, projected_covariances);
  Check(pnec_hip_problem_set_eigensolver_scheme(dev.prob.p, options_.eigensolver_scheme_));
  const Quaterniond q0(initial_pose.rotationMatrix());
  double q[4], t[3];
  Check(pnec_hip_weighted_eigensolver(dev.prob.p, q0.coeffs(), initial_pose.translation().data(),
                                      options_.regularization_, (int32_t)options_.weighted_iterations_, q, t,
                                      PNEC_HIP_MEM_HOST, pnec_hip_frame_stream(dev.frame)));
  return PoseFromQT(q, t);
}

SE3d PNEC::CeresSolver(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                       const std::vector<Matrix3d> &projected_covariances, const SE3d &initial_pose) {
  return CeresSolverFull(bvs1, bvs2, projected_covariances, options_.regularization_, initial_pose);
}

SE3d PNEC::CeresSolverFull(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                           const std::vector<Matrix3d> &projected_covariances, double regularization,
                           const SE3d &initial_pose) {
  // default-constructed optimiser and Target frame, like the reference (pnec.cc:355,366: it ignores
  // Options::ceres_options_ and noise_frame_)
  optimization::PNECCeres optimizer;
  optimizer.InitValues(Quaterniond(initial_pose.rotationMatrix()), initial_pose.translation());
  optimizer.Optimize(bvs1, bvs2, projected_covariances, regularization);
  return optimizer.Result();
}

SE3d PNEC::NECCeresSolver(const bearingVectors_t &bvs1, const bearingVectors_t &bvs2,
                          const SE3d &initial_pose) {
  optimization::NECCeres optimizer;
  optimizer.InitValues(Quaterniond(initial_pose.rotationMatrix()), initial_pose.translation());
  optimizer.Optimize(bvs1, bvs2);
  return optimizer.Result();
}

namespace {
// a ragged batch on the host, flattened the way pnec_hip_problem_fill ingests it
struct FlatBatch {
  std::vector<int64_t> offsets;
  std::vector<Vector3d> b1, b2;
  std::vector<Matrix3d> cv;
  std::vector<double> q0, t0;
  explicit FlatBatch(const std::vector<FramePair> &pairs) {
    const int64_t B = (int64_t)pairs.size();
    offsets.assign(B + 1, 0);
    for (int64_t p = 0; p < B; ++p) {
      if (pairs[p].bvs1.size() != pairs[p].bvs2.size() || pairs[p].bvs1.size() != pairs[p].projected_covs.size())
        throw std::invalid_argument("FramePair arrays differ in size");
      offsets[p + 1] = offsets[p] + (int64_t)pairs[p].bvs1.size();
    }
    b1.reserve(offsets[B]); b2.reserve(offsets[B]); cv.reserve(offsets[B]);
    q0.resize(4 * B); t0.resize(3 * B);
    for (int64_t p = 0; p < B; ++p) {
      b1.insert(b1.end(), pairs[p].bvs1.begin(), pairs[p].bvs1.end());
      b2.insert(b2.end(), pairs[p].bvs2.begin(), pairs[p].bvs2.end());
      cv.insert(cv.end(), pairs[p].projected_covs.begin(), pairs[p].projected_covs.end());
      const Quaterniond q(pairs[p].initial_pose.rotationMatrix());
      std::memcpy(&q0[4 * p], q.coeffs(), 4 * sizeof(double));
      std::memcpy(&t0[3 * p], pairs[p].initial_pose.translation().data(), 3 * sizeof(double));
    }
  }
};
}  // namespace

std::vector<SE3d> PNEC::SolveBatch(const std::vector<FramePair> &pairs, std::vector<std::vector<int>> *inliers) {
  // pnec.cc:77-124 for every pair: one upload, every stage one launch over the batch with the results
  // handed on in HBM (pnec_hip_solve_pipeline), one download
  const int64_t B = (int64_t)pairs.size();
  std::vector<SE3d> out(B);
  if (inliers) inliers->assign(B, {});
  if (B == 0) return out;
  const FlatBatch h(pairs);
  const int64_t total = h.offsets[B];
  const optimization::SolverOptions so;
  Problem prob(so.device, PNEC_HIP_MODE_TARGET, h.offsets);
  if (total > 0)
    Check(pnec_hip_problem_fill(prob.p, 0, B, h.b1[0].data(), h.b2[0].data(), h.cv[0].data(), nullptr,
                                PNEC_HIP_MEM_HOST, nullptr));
  const pnec_hip_pipeline_options po = ToPipeline(options_);
  std::vector<double> q(4 * B), t(3 * B);
  std::vector<uint8_t> mask;
  if (inliers && options_.use_ransac_) mask.assign(total ? total : 1, 0);
  Check(pnec_hip_solve_pipeline(prob.p, h.q0.data(), h.t0.data(), &po, q.data(), t.data(),
                                mask.empty() ? nullptr : mask.data(), nullptr, PNEC_HIP_MEM_HOST, nullptr));
  if (!mask.empty())
    for (int64_t p = 0; p < B; ++p)
      for (int64_t i = h.offsets[p]; i < h.offsets[p + 1]; ++i)
        if (mask[i]) (*inliers)[p].push_back((int)(i - h.offsets[p]));
  for (int64_t p = 0; p < B; ++p) out[p] = PoseFromQT(&q[4 * p], &t[3 * p]);
  return out;
}

std::vector<SE3d> PNEC::SolveBatch(const std::vector<FramePair> &pairs, const std::vector<int> &devices,
                                   std::vector<std::vector<int>> *inliers) {
  const int64_t B = (int64_t)pairs.size();
  std::vector<SE3d> out(B);
  if (inliers) inliers->assign(B, {});
  if (B == 0) return out;
  if (devices.empty()) throw std::invalid_argument("SolveBatch: empty device list");
  const FlatBatch h(pairs);
  const int64_t total = h.offsets[B];
  const pnec_hip_pipeline_options po = ToPipeline(options_);
  std::vector<double> q(4 * B), t(3 * B);
  std::vector<uint8_t> mask;
  if (inliers && options_.use_ransac_) mask.assign(total ? total : 1, 0);
  std::vector<int32_t> devs(devices.begin(), devices.end());
  Check(pnec_hip_solve_pipeline_multi((int32_t)devs.size(), devs.data(), B, h.offsets.data(),
                                      total ? h.b1[0].data() : nullptr, total ? h.b2[0].data() : nullptr,
                                      total ? h.cv[0].data() : nullptr, h.q0.data(), h.t0.data(), &po, q.data(), t.data(),
                                      mask.empty() ? nullptr : mask.data(), nullptr));
  if (!mask.empty())
    for (int64_t p = 0; p < B; ++p)
      for (int64_t i = h.offsets[p]; i < h.offsets[p + 1]; ++i)
        if (mask[i]) (*inliers)[p].push_back((int)(i - h.offsets[p]));
  for (int64_t p = 0; p < B; ++p) out[p] = PoseFromQT(&q[4 * p], &t[3 * p]);
  return out;
}

std::vector<SE3d> PNEC::CeresSolverBatch(const std::vector<FramePair> &pairs,
                                         std::vector<optimization::Summary> *summaries) {
  const int64_t B = (int64_t)pairs.size();
  const FlatBatch h(pairs);
  const std::vector<int64_t> &offsets = h.offsets;
  const auto &b1 = h.b1, &b2 = h.b2;
  const auto &cv = h.cv;
  const auto &q0 = h.q0, &t0 = h.t0;
  const optimization::SolverOptions so;  // defaults, as CeresSolver does
  Problem prob(so.device, PNEC_HIP_MODE_TARGET, offsets);
  if (offsets[B] > 0)
    Check(pnec_hip_problem_fill(prob.p, 0, B, b1[0].data(), b2[0].data(), cv[0].data(), nullptr,
                                PNEC_HIP_MEM_HOST, nullptr));
  std::vector<double> oq(4 * B), ot(3 * B), oc(B);
  std::vector<int32_t> oi(B), os(B);
  const pnec_hip_options o = so.ToHip();
  if (B > 0)
    Check(pnec_hip_solve(prob.p, q0.data(), t0.data(), 1, nullptr, options_.regularization_, &o, oq.data(),
                         ot.data(), oc.data(), oi.data(), os.data(), PNEC_HIP_MEM_HOST, nullptr));
  std::vector<SE3d> out(B);
  if (summaries) summaries->resize(B);
  for (int64_t p = 0; p < B; ++p) {
    out[p] = SE3d(Quaterniond(oq[4 * p + 3], oq[4 * p], oq[4 * p + 1], oq[4 * p + 2]).toRotationMatrix(),
                  Vector3d(ot[3 * p], ot[3 * p + 1], ot[3 * p + 2]));
    if (summaries) (*summaries)[p] = {oc[p], oi[p], os[p]};
  }
  return out;
}

PNEC::~PNEC() {
  if (multi_) pnec_hip_multi_destroy(multi_);
}

std::vector<SE3d> PNEC::CeresSolverBatch(const std::vector<FramePair> &pairs, const std::vector<int> &devices,
                                         std::vector<optimization::Summary> *summaries) {
  const int64_t B = (int64_t)pairs.size();
  std::vector<SE3d> out(B);
  if (summaries) summaries->resize(B);
  if (B == 0) return out;
  if (devices.empty()) throw std::invalid_argument("CeresSolverBatch: empty device list");
  const FlatBatch h(pairs);
  const int64_t total = h.offsets[B];
  int64_t max_pair = 0;
  for (int64_t p = 0; p < B; ++p) max_pair = std::max(max_pair, h.offsets[p + 1] - h.offsets[p]);
  if (!multi_ || multi_devices_ != devices || B > multi_pairs_ || total > multi_corr_ || max_pair > multi_pair_corr_) {
    if (multi_) pnec_hip_multi_destroy(multi_);
    multi_ = nullptr;
    std::vector<int32_t> devs(devices.begin(), devices.end());
    // (some head-room, so that a stream of similar batches keeps the handle)
    multi_pairs_ = B + B / 8; multi_corr_ = total + total / 8; multi_pair_corr_ = max_pair + max_pair / 8;
    Check(pnec_hip_multi_create((int32_t)devs.size(), devs.data(), PNEC_HIP_MODE_TARGET, multi_pairs_, multi_corr_,
                                multi_pair_corr_, &multi_));
    multi_devices_ = devices;
  }
  Check(pnec_hip_multi_fill(multi_, B, h.offsets.data(), total ? h.b1[0].data() : nullptr, total ? h.b2[0].data() : nullptr,
                            total ? h.cv[0].data() : nullptr, nullptr));
  std::vector<double> oq(4 * B), ot(3 * B), oc(B);
  std::vector<int32_t> oi(B), os(B);
  const pnec_hip_options o = optimization::SolverOptions().ToHip();   // defaults, as CeresSolver does (quirk C1)
  Check(pnec_hip_multi_solve(multi_, h.q0.data(), h.t0.data(), 1, nullptr, options_.regularization_, &o, oq.data(), ot.data(),
                             oc.data(), oi.data(), os.data()));
  for (int64_t p = 0; p < B; ++p) {
    out[p] = PoseFromQT(&oq[4 * p], &ot[3 * p]);
    if (summaries) (*summaries)[p] = {oc[p], oi[p], os[p]};
  }
  return out;
}

}  // namespace rel_pose_estimation
}  // namespace pnec
