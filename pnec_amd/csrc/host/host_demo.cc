// host_demo.cc -- compiled into a small executable: run_simulation's call pattern
// (src/run_simulation.cc:74-86) for one synthetic pair through the C++ facade; used by the GPU
// tests to exercise the facade without Python.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "pnec_host.h"

int main(int argc, char **argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 100;
  std::mt19937_64 gen(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::normal_distribution<double> G(0.0, 1.0);
  // ground truth
  const pnec::Quaterniond qgt = pnec::Quaterniond(0.97, 0.1, -0.15, 0.12).normalized();
  const pnec::Matrix3d Rgt = qgt.toRotationMatrix();
  const pnec::Vector3d tgt = pnec::Vector3d(0.3, -0.2, 0.9);
  pnec::bearingVectors_t b1(n), b2(n);
  std::vector<pnec::Matrix3d> covs(n);
  for (int i = 0; i < n; ++i) {
    const double d = 2.0 + 3.0 * U(gen);
    const pnec::Vector3d P((U(gen) - 0.5) * d, (U(gen) - 0.5) * 1.5 * d, d);
    pnec::Vector3d P2 = Rgt.transpose() * (P - tgt);
    const double s = 1e-3;
    P2 = P2 + pnec::Vector3d(s * G(gen), s * G(gen), 0.0) * P2[2];
    b1[i] = P.normalized();
    b2[i] = P2.normalized();
    covs[i] = pnec::Matrix3d::Identity() * (s * s);
  }
  if (argc > 2 && std::strcmp(argv[2], "latency") == 0) {
    // one-pair PNECCeres::Optimize latency, host arrays in, pose out (the reference's per-call pattern:
    // pnec_ceres.cc:70-111 / python/pypnec.cpp:55-65): median and 90th percentile over `reps` calls
    const int reps = argc > 3 ? std::atoi(argv[3]) : 2000;
    const pnec::Quaterniond q0 = pnec::Quaterniond(0.975, 0.1, -0.14, 0.115).normalized();
    const pnec::Vector3d t0 = pnec::Vector3d(0.28, -0.22, 0.92).normalized();
    std::vector<double> us;
    us.reserve((size_t)reps);
    double sink = 0.0;
    int iterations = 0;
    for (int r = 0; r < reps + 50; ++r) {
      pnec::optimization::PNECCeres optimizer;
      optimizer.InitValues(q0, t0);
      const auto tic = std::chrono::steady_clock::now();
      optimizer.Optimize(b1, b2, covs, 1.0e-13);
      const auto toc = std::chrono::steady_clock::now();
      if (r >= 50) us.push_back(std::chrono::duration<double, std::micro>(toc - tic).count());
      sink += optimizer.Translation()[2];
      iterations = optimizer.summary().iterations;
    }
    std::sort(us.begin(), us.end());
    std::printf("{\"call\": \"PNECCeres::Optimize (target frame), host arrays in, pose out\", \"correspondences\": %d, "
                "\"lm_iterations\": %d, \"reps\": %d, \"median_us\": %.2f, \"p10_us\": %.2f, \"p90_us\": %.2f, "
                "\"min_us\": %.2f, \"checksum\": %.6f}\n",
                n, iterations, reps, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], us[0], sink / reps);
    return 0;
  }
  if (argc > 2 && std::strcmp(argv[2], "stream") == 0) {
    // sustained per-frame rate through the C ABI's streaming handle with `window` submits in flight
    const int window = argc > 3 ? std::atoi(argv[3]) : 4, reps = argc > 4 ? std::atoi(argv[4]) : 20000;
    pnec_hip_stream *st = nullptr;
    if (pnec_hip_stream_create(0, n, 1, window, nullptr, &st) != 0) { std::printf("%s\n", pnec_hip_last_error()); return 1; }
    const int64_t offsets[2] = {0, n};
    const pnec::Quaterniond q0 = pnec::Quaterniond(0.975, 0.1, -0.14, 0.115).normalized();
    const pnec::Vector3d t0 = pnec::Vector3d(0.28, -0.22, 0.92).normalized();
    std::vector<int64_t> tickets;
    double q[4], t[3], sink = 0.0;
    auto tic = std::chrono::steady_clock::now();
    for (int r = 0; r < reps + 200; ++r) {
      if (r == 200) tic = std::chrono::steady_clock::now();
      int64_t tk = 0;
      if (pnec_hip_stream_submit(st, PNEC_HIP_MODE_TARGET, 1, offsets, b1[0].data(), b2[0].data(), covs[0].data(), nullptr,
                                 q0.coeffs(), t0.data(), 1.0e-13, nullptr, &tk) != 0) { std::printf("%s\n", pnec_hip_last_error()); return 1; }
      tickets.push_back(tk);
      if ((int)tickets.size() >= window) {
        pnec_hip_stream_wait(st, tickets.front(), q, t, nullptr, nullptr, nullptr);
        tickets.erase(tickets.begin());
        sink += q[3];
      }
    }
    for (int64_t tk : tickets) pnec_hip_stream_wait(st, tk, q, t, nullptr, nullptr, nullptr);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count();
    std::printf("{\"call\": \"pnec_hip_stream_submit/wait, one pair per submit\", \"correspondences\": %d, \"window\": %d, "
                "\"pairs_per_s\": %.1f, \"us_per_pair\": %.2f, \"checksum\": %.6f}\n",
                n, window, reps / secs, secs / reps * 1e6, sink / reps);
    pnec_hip_stream_destroy(st);
    return 0;
  }
  if (argc > 2 && std::strcmp(argv[2], "solve_latency") == 0) {
    // one frame pair per call through the whole default chain, as the odometry front end calls it
    // (frame2frame.cc:129-132): PNEC::Solve(bvs1, bvs2, covs, init, inliers), host arrays in, pose out
    const int reps = argc > 3 ? std::atoi(argv[3]) : 300;
    // optional 4th argument "vo": the options Frame2Frame forces whatever the configuration says -- use_nec, no
    // refinement (frame2frame.cc:127-128)
    const bool vo = argc > 4 && std::strcmp(argv[4], "vo") == 0;
    const bool timed = argc > 4 && std::strcmp(argv[4], "timed") == 0;   // the overload that fills a FrameTiming
    pnec::rel_pose_estimation::Options options;
    if (vo) {
      options.use_nec_ = true;
      options.use_ceres_ = false;
    }
    // optional 5th argument: the eigensolver scheme (include/pnec_hip.h; the facade's default is 2)
    if (argc > 5) options.eigensolver_scheme_ = std::atoi(argv[5]);
    pnec::rel_pose_estimation::PNEC pnec_solver(options);
    const pnec::SE3d init(pnec::Quaterniond(0.975, 0.1, -0.14, 0.115).normalized().toRotationMatrix(),
                          pnec::Vector3d(0.28, -0.22, 0.92).normalized());
    std::vector<double> us;
    double sink = 0.0;
    size_t n_inl = 0;
    pnec::common::FrameTiming last_timing(0);
    for (int r = 0; r < reps + 20; ++r) {
      std::vector<int> inliers;
      pnec::common::FrameTiming frame_timing(r);
      const auto tic = std::chrono::steady_clock::now();
      const pnec::SE3d sol = timed ? pnec_solver.Solve(b1, b2, covs, init, inliers, frame_timing)
                                   : pnec_solver.Solve(b1, b2, covs, init, inliers);
      const auto toc = std::chrono::steady_clock::now();
      last_timing = frame_timing;
      if (r >= 20) us.push_back(std::chrono::duration<double, std::micro>(toc - tic).count());
      sink += sol.translation()[2];
      n_inl = inliers.size();
    }
    std::sort(us.begin(), us.end());
    std::printf("{\"call\": \"PNEC::Solve, %s, "
                "host arrays in, pose + inliers out\", \"eigensolver_scheme\": %d, \"correspondences\": %d, \"inliers\": %zu, \"reps\": %d, "
                "\"median_us\": %.1f, \"p10_us\": %.1f, \"p90_us\": %.1f, \"checksum\": %.6f, \"frame_timing_us\": \"%s\"}\n",
                vo ? "the odometry's forced Options (use_nec, no refinement: RANSAC eigensolver only)"
                   : (timed ? "reference-default Options, the TIMED overload (stage by stage, FrameTiming filled)"
                            : "reference-default Options (RANSAC eigensolver, weighted eigensolver + SCF, refinement)"),
                options.eigensolver_scheme_, n, n_inl, reps, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], sink / reps,
                timed ? last_timing.TimingRowUs().c_str() : "");
    return 0;
  }
  if (argc > 2 && std::strcmp(argv[2], "timing") == 0) {
    // the odometry's bookkeeping around the solver (pnec_vo.cc:220-261,273-276): one FrameTiming per frame,
    // filled by the timed PNEC::Solve overload, collected in a Timing and written to timing.txt
    const int frames = argc > 3 ? std::atoi(argv[3]) : 5;
    const char *path = argc > 4 ? argv[4] : "timing.txt";
    pnec::rel_pose_estimation::Options options;
    pnec::rel_pose_estimation::PNEC pnec_solver(options);
    const pnec::SE3d init(pnec::Quaterniond(0.975, 0.1, -0.14, 0.115).normalized().toRotationMatrix(),
                          pnec::Vector3d(0.28, -0.22, 0.92).normalized());
    pnec::common::Timing timing;
    for (int f = 1; f <= frames; ++f) {
      pnec::common::FrameTiming frame_timing(f);
      std::vector<int> inliers;
      pnec_solver.Solve(b1, b2, covs, init, inliers, frame_timing);
      timing.push_back(frame_timing);
    }
    if (!timing.Save(path)) { std::printf("cannot write %s\n", path); return 1; }
    std::printf("wrote %zu rows to %s\n", timing.size(), path);
    return 0;
  }
  // the reference's default Options: RANSAC eigensolver -> inliers -> 9 weighted eigensolver rounds +
  // SCF -> Ceres-style refinement (run_simulation.cc:74-86 calls it exactly like this)
  pnec::rel_pose_estimation::Options options;
  pnec::rel_pose_estimation::PNEC pnec_solver(options);
  const pnec::SE3d init(pnec::Quaterniond(0.975, 0.1, -0.14, 0.115).normalized().toRotationMatrix(),
                        pnec::Vector3d(0.28, -0.22, 0.92).normalized());
  std::vector<int> inliers;
  const pnec::SE3d sol = pnec_solver.Solve(b1, b2, covs, init, inliers);
  const double e0 = pnec::common::RotationalDifference(init.rotationMatrix(), Rgt);
  const double e1 = pnec::common::RotationalDifference(sol.rotationMatrix(), Rgt);
  const double te = pnec::common::TranslationalDifference(sol.translation(), tgt);
  const double cost = pnec::common::CostFunction(b1, b2, covs, sol);
  std::printf("n=%d inliers=%zu rot_err_init_deg=%.6f rot_err_deg=%.6f t_err_deg=%.6f cost=%.6f\n", n,
              inliers.size(), e0, e1, te, cost);
  if (argc > 3 && std::strcmp(argv[2], "dump") == 0) {
    // everything a checker needs to repeat this call elsewhere: inputs, start pose, result, inliers
    // (text, %.17g: doubles round-trip exactly)
    std::FILE *f = std::fopen(argv[3], "w");
    if (!f) { std::printf("cannot write %s\n", argv[3]); return 1; }
    std::fprintf(f, "%d\n", n);
    for (int i = 0; i < n; ++i) {
      for (int k = 0; k < 3; ++k) std::fprintf(f, "%.17g ", b1[i][k]);
      for (int k = 0; k < 3; ++k) std::fprintf(f, "%.17g ", b2[i][k]);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) std::fprintf(f, "%.17g ", covs[i](r, c));
      std::fprintf(f, "\n");
    }
    const pnec::Quaterniond qi(init.rotationMatrix()), qs(sol.rotationMatrix());
    std::fprintf(f, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", qi.coeffs()[0], qi.coeffs()[1], qi.coeffs()[2],
                 qi.coeffs()[3], init.translation()[0], init.translation()[1], init.translation()[2]);
    std::fprintf(f, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", qs.coeffs()[0], qs.coeffs()[1], qs.coeffs()[2],
                 qs.coeffs()[3], sol.translation()[0], sol.translation()[1], sol.translation()[2]);
    std::fprintf(f, "%zu", inliers.size());
    for (int i : inliers) std::fprintf(f, " %d", i);
    std::fprintf(f, "\n");
    std::fclose(f);
  }
  return (e1 < e0 && e1 < 0.1) ? 0 : 1;
}
