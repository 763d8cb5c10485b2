// pypnec.cpp -- the reference's pybind module surface for the hot path (python/pypnec.cpp:50-82,
// 244-258): pypnec.pyceres / pypnec.pyceresnec with the same call signatures, served by the
// MI355X solver through the host facade.  The KLT / image toy functions of the reference module
// are out of scope (image domain, hard-coded author paths at pypnec.cpp:95-99).
//
//   pyceres(host_bvs, target_bvs, host_covariances, target_covariances, init_pose, regularization) -> 4x4
//   pyceresnec(host_bvs, target_bvs, init_pose) -> 4x4
// host_bvs/target_bvs: sequence of 3-vectors (or an [N,3] array); covariances: sequence of 3x3
// (or [N,3,3]); init_pose: 4x4.  Addition: ceres_solver_batch over ragged lists of pairs.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <vector>

#include "pnec_host.h"

namespace py = pybind11;
using arr = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

std::vector<pnec::Vector3d> ToBearings(const arr &a, const char *name) {
  if (a.ndim() != 2 || a.shape(1) != 3) throw std::invalid_argument(std::string(name) + " must be [N,3]");
  std::vector<pnec::Vector3d> out((size_t)a.shape(0));
  auto r = a.unchecked<2>();
  for (py::ssize_t i = 0; i < a.shape(0); ++i) out[(size_t)i] = pnec::Vector3d(r(i, 0), r(i, 1), r(i, 2));
  return out;
}

std::vector<pnec::Matrix3d> ToCovariances(const arr &a, const char *name) {
  if (a.ndim() != 3 || a.shape(1) != 3 || a.shape(2) != 3)
    throw std::invalid_argument(std::string(name) + " must be [N,3,3]");
  std::vector<pnec::Matrix3d> out((size_t)a.shape(0));
  auto r = a.unchecked<3>();
  for (py::ssize_t i = 0; i < a.shape(0); ++i)
    for (int row = 0; row < 3; ++row)
      for (int col = 0; col < 3; ++col) out[(size_t)i](row, col) = r(i, row, col);
  return out;
}

// Sophus::SE3d(Quaterniond(R).normalized().toRotationMatrix(), t)   (pypnec.cpp:56-59)
pnec::SE3d ToPose(const arr &m) {
  if (m.ndim() != 2 || m.shape(0) != 4 || m.shape(1) != 4) throw std::invalid_argument("init_pose must be 4x4");
  auto r = m.unchecked<2>();
  pnec::Matrix3d R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = r(i, j);
  return pnec::SE3d(pnec::Quaterniond(R).normalized().toRotationMatrix(),
                    pnec::Vector3d(r(0, 3), r(1, 3), r(2, 3)));
}

arr FromPose(const pnec::SE3d &T) {
  arr out({4, 4});
  const pnec::Matrix4d M = T.matrix();
  auto w = out.mutable_unchecked<2>();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) w(i, j) = M[4 * i + j];
  return out;
}

arr pyceres(arr host_bvs, arr target_bvs, arr host_covariances, arr target_covariances, arr init_pose,
            double regularization) {
  const auto b1 = ToBearings(host_bvs, "host_bvs"), b2 = ToBearings(target_bvs, "target_bvs");
  const auto c1 = ToCovariances(host_covariances, "host_covariances");
  const auto c2 = ToCovariances(target_covariances, "target_covariances");
  const pnec::SE3d init = ToPose(init_pose);
  pnec::SE3d result;
  {
    py::gil_scoped_release release;
    pnec::optimization::PNECCeres optimizer;
    optimizer.InitValues(pnec::Quaterniond(init.rotationMatrix()), init.translation());
    optimizer.Optimize(b1, b2, c1, c2, regularization);
    result = optimizer.Result();
  }
  return FromPose(result);
}

arr pyceresnec(arr host_bvs, arr target_bvs, arr init_pose) {
  const auto b1 = ToBearings(host_bvs, "host_bvs"), b2 = ToBearings(target_bvs, "target_bvs");
  const pnec::SE3d init = ToPose(init_pose);
  pnec::SE3d result;
  {
    py::gil_scoped_release release;
    pnec::optimization::NECCeres optimizer;
    optimizer.InitValues(pnec::Quaterniond(init.rotationMatrix()), init.translation());
    optimizer.Optimize(b1, b2);
    result = optimizer.Result();
  }
  return FromPose(result);
}

// PNEC::CeresSolver (target-frame covariances, pnec.cc:350-370) over a list of pairs, one launch.
py::list ceres_solver_batch(py::list bvs1, py::list bvs2, py::list covs, py::list init_poses,
                            double regularization, std::vector<int> devices) {
  const size_t B = bvs1.size();
  if (bvs2.size() != B || covs.size() != B || init_poses.size() != B)
    throw std::invalid_argument("all lists must have one entry per frame pair");
  std::vector<pnec::rel_pose_estimation::FramePair> pairs(B);
  for (size_t p = 0; p < B; ++p) {
    pairs[p].bvs1 = ToBearings(bvs1[p].cast<arr>(), "bvs1[i]");
    pairs[p].bvs2 = ToBearings(bvs2[p].cast<arr>(), "bvs2[i]");
    pairs[p].projected_covs = ToCovariances(covs[p].cast<arr>(), "covs[i]");
    pairs[p].initial_pose = ToPose(init_poses[p].cast<arr>());
  }
  pnec::rel_pose_estimation::Options options;
  options.regularization_ = regularization;
  std::vector<pnec::SE3d> poses;
  {
    py::gil_scoped_release release;
    pnec::rel_pose_estimation::PNEC solver(options);
    poses = devices.empty() ? solver.CeresSolverBatch(pairs) : solver.CeresSolverBatch(pairs, devices);
  }
  py::list out;
  for (const auto &T : poses) out.append(FromPose(T));
  return out;
}

// PNEC::Solve with the reference's default Options (or the flags given) for a list of pairs: every
// stage one launch over the batch.  Returns (poses, inliers).
py::tuple solve_batch(py::list bvs1, py::list bvs2, py::list covs, py::list init_poses, bool use_ransac,
                      bool use_nec, bool use_ceres, int weighted_iterations, double regularization,
                      std::vector<int> devices, int eigensolver_scheme) {
  const size_t B = bvs1.size();
  if (bvs2.size() != B || covs.size() != B || init_poses.size() != B)
    throw std::invalid_argument("all lists must have one entry per frame pair");
  std::vector<pnec::rel_pose_estimation::FramePair> pairs(B);
  for (size_t p = 0; p < B; ++p) {
    pairs[p].bvs1 = ToBearings(bvs1[p].cast<arr>(), "bvs1[i]");
    pairs[p].bvs2 = ToBearings(bvs2[p].cast<arr>(), "bvs2[i]");
    pairs[p].projected_covs = ToCovariances(covs[p].cast<arr>(), "covs[i]");
    pairs[p].initial_pose = ToPose(init_poses[p].cast<arr>());
  }
  pnec::rel_pose_estimation::Options options;
  options.use_ransac_ = use_ransac;
  options.use_nec_ = use_nec;
  options.use_ceres_ = use_ceres;
  options.weighted_iterations_ = (size_t)weighted_iterations;
  options.regularization_ = regularization;
  options.eigensolver_scheme_ = eigensolver_scheme;
  std::vector<pnec::SE3d> poses;
  std::vector<std::vector<int>> inliers;
  {
    py::gil_scoped_release release;
    pnec::rel_pose_estimation::PNEC solver(options);
    poses = devices.empty() ? solver.SolveBatch(pairs, &inliers) : solver.SolveBatch(pairs, devices, &inliers);
  }
  py::list out, inl;
  for (const auto &T : poses) out.append(FromPose(T));
  for (const auto &v : inliers) inl.append(py::cast(v));
  return py::make_tuple(out, inl);
}

// additions used by the tests: the small pnec::common helpers of the facade
arr compose_m(arr bvs1, arr bvs2, arr rotation) {
  const auto b1 = ToBearings(bvs1, "bvs1"), b2 = ToBearings(bvs2, "bvs2");
  auto r = rotation.unchecked<2>();
  pnec::Matrix3d R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R(i, j) = r(i, j);
  const pnec::Matrix3d M = pnec::common::ComposeM(b1, b2, R);
  arr out({3, 3});
  auto w = out.mutable_unchecked<2>();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) w(i, j) = M(i, j);
  return out;
}
arr translation_from_m(arr M) {
  auto r = M.unchecked<2>();
  pnec::Matrix3d Mm;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Mm(i, j) = r(i, j);
  const pnec::Vector3d t = pnec::common::TranslationFromM(Mm);
  arr out({3});
  auto w = out.mutable_unchecked<1>();
  for (int i = 0; i < 3; ++i) w(i) = t[i];
  return out;
}

pnec::Matrix3d ToMatrix3(const arr &a, const char *name) {
  if (a.ndim() != 2 || a.shape(0) != 3 || a.shape(1) != 3) throw std::invalid_argument(std::string(name) + " must be 3x3");
  auto r = a.unchecked<2>();
  pnec::Matrix3d M;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M(i, j) = r(i, j);
  return M;
}
pnec::Vector3d ToVector3(const arr &a, const char *name) {
  if (a.ndim() != 1 || a.shape(0) != 3) throw std::invalid_argument(std::string(name) + " must be a 3-vector");
  auto r = a.unchecked<1>();
  return pnec::Vector3d(r(0), r(1), r(2));
}
// pnec::common metric helpers of the facade (common.cc:210-259), degrees
double rotational_difference(arr R1, arr R2) {
  return pnec::common::RotationalDifference(ToMatrix3(R1, "rotation_1"), ToMatrix3(R2, "rotation_2"));
}
double translational_difference(arr t1, arr t2, bool both_directions) {
  return pnec::common::TranslationalDifference(ToVector3(t1, "translation_1"), ToVector3(t2, "translation_2"),
                                               both_directions);
}
double cost_function(arr bvs1, arr bvs2, arr covs, arr pose) {
  return pnec::common::CostFunction(ToBearings(bvs1, "bvs_1"), ToBearings(bvs2, "bvs_2"),
                                    ToCovariances(covs, "covs"), ToPose(pose));
}

// PNEC::Solve for ONE frame pair through the overload asked for (pnec.cc:69-75, :77-124, :126-134,
// :135-208): overload 0 = (bvs1, bvs2, covs, init), 1 = (+ inliers), 2 = (+ timing), 3 = (+ inliers,
// timing).  Returns (pose 4x4, inliers or None, timing dict or None).
py::tuple solve(arr bvs1, arr bvs2, arr covs, arr init_pose, int overload, bool use_ransac, bool use_nec,
                bool use_ceres, int weighted_iterations, double regularization, int eigensolver_scheme,
                bool ransac_chained_starts) {
  const auto b1 = ToBearings(bvs1, "bvs1"), b2 = ToBearings(bvs2, "bvs2");
  const auto cv = ToCovariances(covs, "covs");
  const pnec::SE3d init = ToPose(init_pose);
  pnec::rel_pose_estimation::Options options;
  options.use_ransac_ = use_ransac;
  options.use_nec_ = use_nec;
  options.use_ceres_ = use_ceres;
  options.weighted_iterations_ = (size_t)weighted_iterations;
  options.regularization_ = regularization;
  options.eigensolver_scheme_ = eigensolver_scheme;
  options.ransac_chained_starts_ = ransac_chained_starts;
  if (overload < 0 || overload > 3) throw std::invalid_argument("overload must be 0..3");
  pnec::SE3d pose;
  std::vector<int> inliers;
  pnec::common::FrameTiming timing(7);
  // sentinels: the timed overloads only write the fields the reference writes (pnec.cc:135-208)
  timing.nec_es_ = timing.it_es_ = timing.avg_it_es_ = timing.ceres_ = -1;
  {
    py::gil_scoped_release release;
    pnec::rel_pose_estimation::PNEC solver(options);
    switch (overload) {
      case 0: pose = solver.Solve(b1, b2, cv, init); break;
      case 1: pose = solver.Solve(b1, b2, cv, init, inliers); break;
      case 2: pose = solver.Solve(b1, b2, cv, init, timing); break;
      default: pose = solver.Solve(b1, b2, cv, init, inliers, timing); break;
    }
  }
  py::object inl = py::none(), tim = py::none();
  if (overload == 1 || overload == 3) inl = py::cast(inliers);
  if (overload >= 2) {
    py::dict d;
    d["id"] = timing.id_;
    d["nec_es"] = timing.nec_es_;
    d["it_es"] = timing.it_es_;
    d["avg_it_es"] = timing.avg_it_es_;
    d["ceres"] = timing.ceres_;
    d["frame_loading"] = timing.frame_loading_;
    d["feature_creation"] = timing.feature_creation_;
    d["optimization"] = timing.OptimizationTime();
    d["total"] = timing.TotalTime();
    d["header"] = pnec::common::FrameTiming::TimingHeader();
    // the microsecond twins (this library's addition: a whole Solve is a fraction of a millisecond here)
    d["nec_es_us"] = timing.nec_es_us_;
    d["it_es_us"] = timing.it_es_us_;
    d["avg_it_es_us"] = timing.avg_it_es_us_;
    d["ceres_us"] = timing.ceres_us_;
    d["optimization_us"] = timing.OptimizationTimeUs();
    d["row_us"] = timing.TimingRowUs();
    d["header_us"] = pnec::common::FrameTiming::TimingHeaderUs();
    tim = d;
  }
  return py::make_tuple(FromPose(pose), inl, tim);
}

int add(int i, int j) { return i + j; }  // the reference module's smoke function (pypnec.cpp:34)

}  // namespace

PYBIND11_MODULE(pypnec, m) {
  m.doc() = "PNEC least-squares refinement on AMD MI355X (drop-in for tum-vision/pnec's pypnec)";
  m.def("add", &add, "A function that adds two numbers");
  m.def("pyceres", &pyceres, py::arg("host_bvs"), py::arg("target_bvs"), py::arg("host_covariances"),
        py::arg("target_covariances"), py::arg("init_pose"), py::arg("regularization"),
        "Symmetric PNEC refinement (PNECCeres::Optimize with covariances in both frames)");
  m.def("pyceresnec", &pyceresnec, py::arg("host_bvs"), py::arg("target_bvs"), py::arg("init_pose"),
        "NEC refinement (NECCeres::Optimize)");
  m.def("compose_m", &compose_m, "pnec::common::ComposeM (loop from i = 1, like the reference)");
  m.def("translation_from_m", &translation_from_m, "pnec::common::TranslationFromM");
  m.def("rotational_difference", &rotational_difference, "pnec::common::RotationalDifference (degrees)");
  m.def("translational_difference", &translational_difference, py::arg("translation_1"), py::arg("translation_2"),
        py::arg("both_directions") = true, "pnec::common::TranslationalDifference (degrees)");
  m.def("cost_function", &cost_function, "pnec::common::CostFunction (device)");
  m.def("solve", &solve, py::arg("bvs1"), py::arg("bvs2"), py::arg("covs"), py::arg("init_pose"),
        py::arg("overload") = 1, py::arg("use_ransac") = true, py::arg("use_nec") = false,
        py::arg("use_ceres") = true, py::arg("weighted_iterations") = 10, py::arg("regularization") = 1e-13,
        py::arg("eigensolver_scheme") = 2, py::arg("ransac_chained_starts") = false,
        "PNEC::Solve for one frame pair through one of its four overloads (eigensolver_scheme: which iteration stands in "
        "for opengv's eigenvalue minimisation; ransac_chained_starts: PNEC_HIP_RANSAC_CHAINED_STARTS -- include/pnec_hip.h)");
  m.def("solve_batch", &solve_batch, py::arg("bvs1"), py::arg("bvs2"), py::arg("covs"), py::arg("init_poses"),
        py::arg("use_ransac") = true, py::arg("use_nec") = false, py::arg("use_ceres") = true,
        py::arg("weighted_iterations") = 10, py::arg("regularization") = 1e-13, py::arg("devices") = std::vector<int>{},
        py::arg("eigensolver_scheme") = 2,
        "PNEC::Solve for a list of frame pairs, every stage one device launch over the batch (addition); devices: the "
        "GPUs to shard the pairs over from this process (empty: the default device)");
  m.def("ceres_solver_batch", &ceres_solver_batch, py::arg("bvs1"), py::arg("bvs2"), py::arg("covs"),
        py::arg("init_poses"), py::arg("regularization") = 1e-13, py::arg("devices") = std::vector<int>{},
        "PNEC::CeresSolver for a list of frame pairs in one device launch (addition); devices: the GPUs to shard the pairs "
        "over from this process (empty: the default device)");
}
