// pnec_solve_kernel.hpp -- the persistent on-device Levenberg-Marquardt kernel.
//
// One workgroup = one solve (pair x hypothesis).  WPP wavefronts cooperate; each lane keeps CPL
// correspondences (12 doubles each for TARGET) in registers for the whole loop, so the batch is
// read from HBM exactly once and the iteration loop never leaves the device.
//
// The loop restates Ceres' TrustRegionMinimizer + LevenbergMarquardtStrategy with the options the
// reference's default-constructed optimiser uses (src/optimization/pnec_ceres.cc:47,110;
// SURVEY.md Appendix B), on the reference's parameterisation: theta, phi Euclidean,
// quaternion with EigenQuaternionManifold (pnec_ceres.cc:98-106).  Differences, by design:
//   * closed-form Jacobian instead of central differences (same derivative, no 13x functor cost);
//   * cost, J'J and J'r of the CANDIDATE point are produced by one fused pass, so an accepted step
//     needs no second evaluation (Ceres evaluates residuals at the candidate, then residuals +
//     Jacobian again after accepting).  Same numbers, half the passes.
#pragma once

#include "pnec_device.hpp"

// (correspondences per lane, wavefronts per solve) geometries that are instantiated for every
// residual family.  capacity = 64 * CPL * WPP correspondences held in registers.
#define PNEC_FOR_EACH_GEOMETRY(X) \
  X(1, 1) X(2, 1) X(4, 1) X(8, 1) \
  X(4, 2) X(2, 4) X(1, 8) X(4, 4) X(4, 8)
constexpr int kStreamWaves = 8;  // block shape of the streaming (non-resident) fallback

namespace pnec_hip {

struct SolveArgs {
  const double *data;           // SoA payload
  const int64_t *block_offset;  // [n_pairs] first double of the pair's block
  const int32_t *count;         // [n_pairs] correspondences of the pair
  const double *init_q;         // [n_pairs,4]
  const double *init_t;         // [n_pairs,3]
  const double *hyp_t;          // [n_solves,3] or null
  double *out_q;                // [n_solves,4] or null
  double *out_t;                // [n_solves,3] or null
  double *out_cost;             // [n_solves] or null
  int32_t *out_iterations;      // [n_solves] or null
  int32_t *out_status;          // [n_solves] or null
  int64_t n_solves;
  int32_t n_hyp;
  double reg;
  pnec_hip_options opt;
};

// Contiguous chunks of solves per XCD: block b runs on XCD b%8 (observed dispatch order), so the
// hypotheses of one pair -- consecutive solve indices reading the same payload -- share one L2.
// Bijective for any n (cdna guide T1).  Placement only affects speed.
__device__ __forceinline__ int64_t xcd_contiguous_index(int64_t b, int64_t n) {
  const int64_t xcd = b & 7, q = n >> 3, r = n & 7;
  const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

template <int MODE, int CPL, int WPP, bool RESIDENT>
__global__ __launch_bounds__(kWave *WPP) void lm_solve_kernel(const SolveArgs a) {
  constexpr int NC = num_components(MODE);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  const int64_t s = xcd_contiguous_index(blockIdx.x, a.n_solves);
  const int64_t pair = s / a.n_hyp;
  const double *__restrict__ base = a.data + a.block_offset[pair];
  const int n = a.count[pair];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const pnec_hip_options &o = a.opt;
  const double reg = a.reg;

  // ---- load this lane's correspondences once (coalesced: consecutive lanes, consecutive doubles)
  double d[RESIDENT ? CPL : 1][NC];
  unsigned vmask = 0;
  if constexpr (RESIDENT) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int idx = (wave * CPL + k) * kWave + lane;
      const bool in = idx < stride;
#pragma unroll
      for (int c = 0; c < NC; ++c) d[k][c] = in ? base[(int64_t)c * stride + idx] : 0.0;
      vmask |= (idx < n ? 1u : 0u) << k;
    }
  }

  [[maybe_unused]] __shared__ double xw[2][WPP > 1 ? WPP : 1][kNumAcc];
  int parity = 0;

  // one pass over the pair: all-reduced sums, identical bits in every lane of every wave
  auto run_pass = [&](const PassUniforms &U, double(&sum)[kNumAcc]) {
    double acc[kNumAcc];
#pragma unroll
    for (int j = 0; j < kNumAcc; ++j) acc[j] = 0.0;
    if constexpr (RESIDENT) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        double r, J[5];
        eval_corr<MODE>(d[k], (vmask >> k) & 1u, U, reg, r, J);
        accumulate(r, J, acc);
      }
    } else {
      for (int idx = threadIdx.x; idx < stride; idx += kWave * WPP) {
        double e[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) e[c] = base[(int64_t)c * stride + idx];
        double r, J[5];
        eval_corr<MODE>(e, idx < n, U, reg, r, J);
        accumulate(r, J, acc);
      }
    }
#pragma unroll
    for (int j = 0; j < kNumAcc; ++j) acc[j] = wave_allreduce_sum(acc[j]);
    if constexpr (WPP > 1) {
      if (lane < kNumAcc) {
        // lane j publishes sum j (all lanes hold all sums; pick by lane without dynamic indexing)
        double v = acc[0];
#pragma unroll
        for (int j = 1; j < kNumAcc; ++j) v = (lane == j) ? acc[j] : v;
        xw[parity][wave][lane] = v;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < kNumAcc; ++j) {
        double t = xw[parity][0][j];
#pragma unroll
        for (int w = 1; w < WPP; ++w) t += xw[parity][w][j];
        acc[j] = t;
      }
      parity ^= 1;
    }
#pragma unroll
    for (int j = 0; j < kNumAcc; ++j) sum[j] = to_sgpr(acc[j]);
  };

  // ---- PNECCeres::InitValues(q, t): pnec_ceres.cc:182-186
  double q[4], theta, phi;
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = to_sgpr(a.init_q[pair * 4 + k]);
  {
    const double *t0 = a.hyp_t ? a.hyp_t + 3 * s : a.init_t + 3 * pair;
    angles_from_vec(t0[0], t0[1], t0[2], theta, phi);
    theta = to_sgpr(theta);
    phi = to_sgpr(phi);
  }

  double cost, Hs[15], gs[5], scale[5], diag[5], gmax;
  int iteration = 0, term = PNEC_HIP_TERM_MAX_ITERATIONS;

  // sums -> cost, scaled H and g (delta = omega/2 => omega columns x2)
  auto unpack = [&](const double(&S)[kNumAcc], double &c, double(&H)[15], double(&g)[5]) {
    c = 0.5 * S[0];
#pragma unroll
    for (int i = 0; i < 5; ++i) g[i] = S[1 + i] * (i >= 2 ? 2.0 : 1.0);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j)
        H[tri(i, j)] = S[6 + tri(i, j)] * ((i >= 2 ? 2.0 : 1.0) * (j >= 2 ? 2.0 : 1.0));
  };
  auto all_finite = [&](double c, const double(&H)[15], const double(&g)[5]) {
    bool ok = finite_d(c);
#pragma unroll
    for (int i = 0; i < 5; ++i) ok = ok && finite_d(g[i]);
#pragma unroll
    for (int i = 0; i < 15; ++i) ok = ok && finite_d(H[i]);
    return ok;
  };
  auto rescale = [&](const double(&H)[15], const double(&g)[5]) {
    gmax = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      gs[i] = to_sgpr(g[i] * scale[i]);
      gmax = fmax(gmax, fabs(g[i]));
    }
    gmax = to_sgpr(gmax);
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = i; j < 5; ++j) Hs[tri(i, j)] = to_sgpr(H[tri(i, j)] * (scale[i] * scale[j]));
  };

  {  // ---- iteration zero
    PassUniforms U;
    make_uniforms(theta, phi, q, U);
    double S[kNumAcc], H[15], g[5];
    run_pass(U, S);
    unpack(S, cost, H, g);
    cost = to_sgpr(cost);
    if (to_sgpr((int)all_finite(cost, H, g)) == 0) {
      term = PNEC_HIP_TERM_BAD_INITIAL;
      goto finish;
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
      scale[i] = to_sgpr(o.jacobi_scaling ? 1.0 / (1.0 + sqrt(H[tri(i, i)])) : 1.0);
    rescale(H, g);
  }

  {
    double x_norm = to_sgpr(sqrt(theta * theta + phi * phi + q[0] * q[0] + q[1] * q[1] +
                                 q[2] * q[2] + q[3] * q[3]));
    double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
    int reuse_diagonal = 0, num_invalid = 0, step_ok = 1;
#pragma unroll
    for (int i = 0; i < 5; ++i) diag[i] = 0.0;

    for (;;) {
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      if (iteration >= o.max_num_iterations) { term = PNEC_HIP_TERM_MAX_ITERATIONS; break; }
      if (to_sgpr((int)(o.check_convergence && step_ok && gmax <= o.gradient_tolerance))) {
        term = PNEC_HIP_TERM_GRADIENT_TOL; break;
      }
      if (to_sgpr((int)(radius < o.min_trust_region_radius))) {
        term = PNEC_HIP_TERM_MIN_RADIUS; break;
      }
      ++iteration;
      step_ok = 0;

      // LevenbergMarquardtStrategy::ComputeStep
      if (!reuse_diagonal) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          diag[i] = to_sgpr(fmin(fmax(Hs[tri(i, i)], o.min_lm_diagonal), o.max_lm_diagonal));
      }
      double A[15], y[5], step[5];
      const double inv_radius = fast_rcp(radius);
#pragma unroll
      for (int i = 0; i < 15; ++i) A[i] = Hs[i];
#pragma unroll
      for (int i = 0; i < 5; ++i) A[tri(i, i)] = __builtin_fma(diag[i], inv_radius, A[tri(i, i)]);
      bool valid = chol_solve5(A, gs, y);
      double model_change = 0.0;
      {
        double sg = 0.0, shs = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) step[i] = -y[i];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          sg = __builtin_fma(step[i], gs[i], sg);
          double row = 0.0;
#pragma unroll
          for (int j = 0; j < 5; ++j) row = __builtin_fma(Hs[sym(i, j)], step[j], row);
          shs = __builtin_fma(step[i], row, shs);
        }
        model_change = -(sg + 0.5 * shs);
        valid = valid && (model_change > 0.0);
      }
      if (to_sgpr((int)valid) == 0) {
        if (++num_invalid >= o.max_num_consecutive_invalid_steps) {
          term = PNEC_HIP_TERM_INVALID_STEPS; break;
        }
        radius = to_sgpr(radius / decrease_factor);
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
        continue;
      }
      num_invalid = 0;

      // candidate = Plus(x, step * jacobi_scale): EigenQuaternionManifold::Plus on q
      double qc[4];
      const double thc = to_sgpr(theta + step[0] * scale[0]);
      const double phc = to_sgpr(phi + step[1] * scale[1]);
      {
        const double dx = step[2] * scale[2], dy = step[3] * scale[3], dz = step[4] * scale[4];
        const double nd = sqrt(dx * dx + dy * dy + dz * dz);
        if (nd == 0.0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) qc[k] = q[k];
        } else {
          double sn, aw;
          sincos(nd, &sn, &aw);
          const double sbd = sn / nd;
          const double ax = sbd * dx, ay = sbd * dy, az = sbd * dz;
          qc[0] = aw * q[0] + ax * q[3] + ay * q[2] - az * q[1];
          qc[1] = aw * q[1] - ax * q[2] + ay * q[3] + az * q[0];
          qc[2] = aw * q[2] + ax * q[1] - ay * q[0] + az * q[3];
          qc[3] = aw * q[3] - ax * q[0] - ay * q[1] - az * q[2];
        }
      }

#pragma unroll
      for (int k = 0; k < 4; ++k) qc[k] = to_sgpr(qc[k]);
      model_change = to_sgpr(model_change);
      // one fused pass at the candidate: cost, J'J, J'r
      PassUniforms U;
      make_uniforms(thc, phc, qc, U);
      double S[kNumAcc], Hc[15], gc[5], cost_c;
      run_pass(U, S);
      unpack(S, cost_c, Hc, gc);
      const bool cand_ok = all_finite(cost_c, Hc, gc);
      if (!finite_d(cost_c)) cost_c = 1.7976931348623157e308;
      cost_c = to_sgpr(cost_c);

      if (o.check_convergence) {
        const double dq0 = q[0] - qc[0], dq1 = q[1] - qc[1], dq2 = q[2] - qc[2],
                     dq3 = q[3] - qc[3];
        const double step_norm = sqrt((theta - thc) * (theta - thc) + (phi - phc) * (phi - phc) +
                                      dq0 * dq0 + dq1 * dq1 + dq2 * dq2 + dq3 * dq3);
        if (to_sgpr((int)(step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)))) {
          term = PNEC_HIP_TERM_PARAMETER_TOL; break;
        }
        if (to_sgpr((int)(fabs(cost - cost_c) <= o.function_tolerance * cost))) {
          term = PNEC_HIP_TERM_FUNCTION_TOL; break;
        }
      }
      const double rho = (cost - cost_c) / model_change;
      if (to_sgpr((int)(rho > o.min_relative_decrease))) {
        if (to_sgpr((int)cand_ok) == 0) {  // finite cost but non-finite Jacobian: Ceres fails here
          term = PNEC_HIP_TERM_BAD_INITIAL; break;
        }
        theta = thc;
        phi = phc;
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = qc[k];
        x_norm = to_sgpr(sqrt(theta * theta + phi * phi + q[0] * q[0] + q[1] * q[1] +
                              q[2] * q[2] + q[3] * q[3]));
        cost = cost_c;
        rescale(Hc, gc);
        step_ok = 1;
        const double c1 = 2.0 * rho - 1.0;
        radius = radius / fmax(1.0 / 3.0, 1.0 - c1 * c1 * c1);
        radius = to_sgpr(fmin(o.max_trust_region_radius, radius));
        decrease_factor = 2.0;
        reuse_diagonal = 0;
      } else {
        radius = to_sgpr(radius / decrease_factor);
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
      }
    }
  }

finish:
  // ---- PNECCeres::Result(): pnec_ceres.cc:201-207
  if (threadIdx.x == 0) {
    const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (a.out_q) {
#pragma unroll
      for (int k = 0; k < 4; ++k) a.out_q[4 * s + k] = q[k] * qn;
    }
    if (a.out_t) {
      double st, ct, sp, cp;
      sincos(theta, &st, &ct);
      sincos(phi, &sp, &cp);
      a.out_t[3 * s + 0] = st * cp;
      a.out_t[3 * s + 1] = st * sp;
      a.out_t[3 * s + 2] = ct;
    }
    if (a.out_cost) a.out_cost[s] = cost;
    if (a.out_iterations) a.out_iterations[s] = iteration;
    if (a.out_status) a.out_status[s] = term;
  }
}

}  // namespace pnec_hip
