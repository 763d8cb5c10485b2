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
//
// Register / LDS plan (what makes one wavefront per 512-correspondence pair viable):
//   * the pair's payload lives in VGPRs (+ LDS slots, + AGPRs when the compiler needs room) for
//     the whole loop; the 17 pose uniforms of a pass live in SGPRs;
//   * a pass ends with the 21 sums in a per-wavefront LDS slab (swap-halving + DPP
//     reduction, pnec_device.hpp);
//   * everything that is one value per solve -- accept/reject, trust region, the 5x5 solve, the
//     manifold update, the next pose's uniforms (lm_advance) -- runs in one quad on LDS-resident
//     state (~80 doubles per wavefront): a ~500-instruction latency chain that the SIMD's other
//     wavefront hides under its pass.  (Measured dead ends, DESIGN.md section 6: narrowing EXEC
//     does not shorten the issue time of that chain, and sharing one chain between the
//     wavefronts of a workgroup trades its issue slots for barrier stalls -- no gain.)
#pragma once

#include <type_traits>

#include "pnec_device.hpp"


// Launch geometries instantiated for every residual family:
//   (CPL correspondences per lane, WPP wavefronts per solve, LDSK of the CPL kept in LDS)
// capacity = 64 * CPL * WPP correspondences resident on chip for the whole LM loop.
// LDSK > 0 moves that many of a lane's correspondences from registers to LDS so that a wavefront
// fits 256 registers and two wavefronts share a SIMD (latency hiding) while still owning 512
// correspondences each -- the serial part of an iteration is paid once per wavefront, so fat
// wavefronts win.  (8,W,3) covers N = 512 W up to 4096 (5 correspondences per lane in registers,
// 3 in LDS: 18.4 KB per wavefront, 8 wavefronts per CU); (4,W,0) is the family for the 18-plane
// SYM payload; (8,1,0), (4,2,0) and (1,8,0) also serve the A/B measurements in DESIGN.md.
#define PNEC_FOR_EACH_GEOMETRY(X) \
  X(1, 1, 0) X(2, 1, 0) X(4, 1, 0) X(4, 2, 0) X(4, 4, 0) X(4, 8, 0) \
  X(8, 1, 3) X(12, 1, 3) X(8, 2, 3) X(8, 4, 3) X(8, 8, 3) X(8, 1, 0) X(1, 8, 0)
// (12, 1, 3) is (8, 1, 3) + a TAIL: 512 correspondences resident on chip and up to 256 more re-read from L2 in every
// pass -- pairs of 513..768 on ONE wavefront instead of (8, 2, 3)'s two wavefronts and two barriers per pass for a
// tail of a few dozen correspondences.  See the tail pass in lm_solve_kernel: its sums are, bit for bit, those of
// (8, 2, 3)'s second wavefront.
// the geometries the auto-tuner's ladders can pick (pnec_capi.hip geometry_ladder): what the AoS-source
// kernels of the streaming handle are built for
#define PNEC_FOR_EACH_AOS_GEOMETRY(X) \
  X(1, 1, 0) X(2, 1, 0) X(4, 1, 0) X(4, 4, 0) X(4, 8, 0) X(8, 1, 3) X(8, 2, 3) X(8, 4, 3) X(8, 8, 3) X(8, 1, 0)
constexpr int kStreamWaves = 8;  // block shape of the streaming (non-resident) fallback

namespace pnec_hip {

struct SolveArgs {
  const double *data;           // SoA payload
  const int64_t *block_offset;  // [n_pairs] first double of the pair's block
  const int32_t *count;         // [n_pairs] correspondences of the pair
  const int32_t *pair_index;    // null, or the pairs this launch covers (ragged batches: one launch
                                // per geometry); solve slot i of the launch = pair_index[i / n_hyp]
  const double *init_q;         // [n_pairs,4]
  const double *init_t;         // [n_pairs,3]
  const double *hyp_t;          // [n_solves,3] or null
  double *out_q;                // [n_solves,4] or null
  double *out_t;                // [n_solves,3] or null
  double *out_cost;             // [n_solves] or null
  int32_t *out_iterations;      // [n_solves] or null
  int32_t *out_status;          // [n_solves] or null
  unsigned long long *trace;    // null, or [n_blocks,4]: s_memtime at start / payload on chip / end, hw id
  unsigned long long *work;     // null, or [2]: correspondence-passes executed in full / cost-only (PNEC_HIP_OPT_COUNT_PASSES)
  int64_t n_solves;
  int32_t n_hyp;
  int32_t numeric_jacobian;     // PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL (verification; streaming form only)
  // SRC_AOS (streaming handle, pnec_stream.hip): the pairs are read straight from the caller's arrays in
  // the REFERENCE layout (bvs 3 doubles, covs 9 doubles column-major per correspondence) -- pinned host
  // memory mapped into the device, so one launch ingests, solves and reports without a staging copy
  const double *aos_bvs1, *aos_bvs2, *aos_covs, *aos_covs_host;
  const int64_t *aos_offsets;            // [n_pairs + 1] correspondence offsets of the submit
  unsigned long long *done_counter;      // device: blocks of this submit that have written their result
  unsigned long long *host_flag;         // pinned host: receives flag_value when all n_blocks_total are done
  unsigned long long flag_value, n_blocks_total;
  double reg;
  double inv_max_radius;  // 1 / opt.max_trust_region_radius, 1 / opt.min_trust_region_radius: the loop
  double inv_min_radius;  // carries the inverse radius (lm_advance); filled by the host (finish_args)
  pnec_hip_options opt;
};
inline void finish_args(SolveArgs &a) {
  a.inv_max_radius = 1.0 / a.opt.max_trust_region_radius;
  a.inv_min_radius = 1.0 / a.opt.min_trust_region_radius;
}

// Contiguous chunks of solves per XCD: block b runs on XCD b%8 (observed dispatch order), so the
// hypotheses of one pair -- consecutive solve indices reading the same payload -- share one L2.
// Bijective for any n (cdna guide T1).  Placement only affects speed.
__device__ __forceinline__ int64_t xcd_contiguous_index(int64_t b, int64_t n) {
  const int64_t xcd = b & 7, q = n >> 3, r = n & 7;
  const int64_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// slots of the per-wavefront LDS slab (doubles)
enum : int {
  kQ = 0,        // 4  current quaternion
  kTheta = 4,    // 1
  kPhi = 5,      // 1
  kCost = 6,     // 1
  kXNorm = 7,    // 1
  kInvRadius = 8,  // 1  1 / trust-region radius (the radius itself is never needed: see lm_advance)
  kDec = 9,      // 1  decrease_factor (a power of two)
  kModel = 10,   // 1  model cost change of the pending step
  kQc = 11,      // 4  candidate quaternion
  kThetaC = 15,  // 1
  kPhiC = 16,    // 1
  kDiag = 17,    // 5  LM diagonal, mapped to the pass's parameter scale (see the step below)
  kScaleSq = 22, // 5  (Jacobi scale x (2 for the rotation columns))^2
  kInvScaleSq = 27,  // 5
  kGmax = 32,    // 1
  kSumsFinite = 33,  // 1  1.0 when all 21 sums are finite
  // two tables of the pass's 21 sums as the row leaders store them (sum_slot): one holds the normal
  // equations of the current (accepted) point, the pass writes the candidate's into the other; accepting
  // a step flips which is which (ist[kIPark]) instead of copying 20 doubles
  kSums = 36,    // 2 x 24
  kTcur = 84,    // 3  translation of the current point: the candidate's pass uniforms t = unif[9..11], copied when the
                 //    candidate becomes the current point, so Result() needs no sine / cosine of (theta, phi) again
  kSlab = 88
};
constexpr int kUnif = 18;   // pass uniforms of the candidate: R[9] | t[3] | dt/dtheta[3] | dt/dphi[2] | pad
// cost-only pass after a rejected step (lm_advance<COST_FIRST>); -DPNEC_NO_COST_FIRST: the always-speculating kernel (A/B)
#ifdef PNEC_NO_COST_FIRST
constexpr bool kCostFirst = false;
#else
constexpr bool kCostFirst = true;
#endif
// per-solve integer state (LDS)
enum : int { kIIter = 0, kIFirst, kIReuseDiag, kINumInvalid, kIStepOk,
              kILast,  // the published candidate is evaluated at the iteration cap: its Jacobian can never be used
              kIPark,  // which table of sums (0 / 1) belongs to the current point
              kITerm,  // several wavefronts per solve: the termination code the advancing wavefront publishes
              kINumI = 8 };

// Which (family, geometry) pairs are built: the payload must fit the 160 KB LDS and the
// register budget of its occupancy target (<= 72 doubles of payload per lane at two wavefronts
// per SIMD; the (8,W,0) shape runs one wavefront per SIMD with AGPR parking).
__host__ __device__ constexpr bool geometry_ok(int mode, int cpl, int wpp, int ldsk) {
  const int nc = num_components(mode);
  if (cpl == 12) return wpp == 1 && ldsk == 3 && nc <= 12;  // (8, 1, 3) + tail: the 6- and 12-plane payloads
#ifdef PNEC_ADVANCE_ROWS
  constexpr long ab_lds = 16 * 8 * 4;  // gather offsets of lm_advance_rows (A/B build only)
#else
  constexpr long ab_lds = 0;
#endif
  const long lds = (long)wpp * (ldsk * nc * kWave * 8 + (kSlab + kUnif) * 8 + kINumI * 4) + (wpp > 1 ? 2L * wpp * kSumSlots * 8 : 0) + ab_lds;
  if (lds > 160 * 1024) return false;
  if (cpl == 8 && ldsk == 0) return nc <= 18;
  return nc * (cpl - ldsk) <= 72;
}


// ---- shared by the solve kernels -------------------------------------------------------------
// Which correspondence a lane's slot k holds (relative to the wavefront's first): the REGK register
// slots come in pairs (2j, 2j+1) <-> 128 j + 2 lane + {0, 1} -- two neighbouring correspondences per
// 16-byte load, the CU's load path moving ~2x the bytes per clock of 8-byte loads -- a leftover odd
// register slot and the LDS slots hold 64 consecutive correspondences each (lane l <-> chunk + l).
// Monotone in k for lane 0, so "slot k is empty for every lane" is a prefix property (nslots).
template <int CPL, int REGK>
__host__ __device__ constexpr int slot_corr(int k, int lane) {
  if (CPL == 1) return lane;
  if (k < (REGK & ~1)) return 2 * kWave * (k / 2) + 2 * lane + (k & 1);
  return kWave * k + lane;  // the odd register slot (k = REGK - 1) and the LDS slots
}

// Load one wavefront's share of a pair (CPL correspondences per lane starting at first_corr) into
// REGK register slots + (CPL - REGK) LDS slots.  The LDS slots are filled by the DMA path
// (global_load_lds_dwordx4: memory -> LDS without passing through VGPRs): with the register file full
// of payload there are no temporaries to stage them through, and staging them a few registers at a
// time exposed the HBM latency once per batch (the load phase was 11 % of a wavefront's life).  One
// instruction copies 512 B = one plane of one slot (lanes 0..31, 16 B each, landing linearly).
template <int NC, int CPL, int REGK>
__device__ __forceinline__ void load_resident(const double *__restrict__ base, int n, int stride,
                                              int first_corr, int lane, double (&d)[REGK][NC],
                                              double *lds /* [CPL-REGK][NC][64] */) {
  (void)n;
  if constexpr (CPL == 1) {
    const int idx = first_corr + lane;
    const bool in = idx < stride;
#pragma unroll
    for (int c = 0; c < NC; ++c) d[0][c] = in ? base[(int64_t)c * stride + idx] : 0.0;
  } else {
    using pair_t = __attribute__((ext_vector_type(2))) double;
    auto load_pair = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int idx = first_corr + 2 * kWave * j + 2 * lane;
      const bool in = idx < stride;  // stride is a multiple of 64, idx is even: idx+1 < stride too
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        pair_t v = {0.0, 0.0};
        if (in) v = *reinterpret_cast<const pair_t *>(base + (int64_t)c * stride + idx);
        d[2 * j][c] = v.x;
        d[2 * j + 1][c] = v.y;
      }
    };
    if constexpr (REGK >= 2) load_pair(std::integral_constant<int, 0>{});
    if constexpr (REGK >= 4) load_pair(std::integral_constant<int, 1>{});
    if constexpr (REGK >= 6) load_pair(std::integral_constant<int, 2>{});
    if constexpr (REGK >= 8) load_pair(std::integral_constant<int, 3>{});
    if constexpr (REGK & 1) {
      const int idx = first_corr + kWave * (REGK - 1) + lane;
      const bool in = idx < stride;
#pragma unroll
      for (int c = 0; c < NC; ++c) d[REGK - 1][c] = in ? base[(int64_t)c * stride + idx] : 0.0;
    }
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void global_void;
#pragma unroll
    for (int k = REGK; k < CPL; ++k) {
      const int chunk = first_corr + kWave * k;  // 64 consecutive correspondences, 64-aligned
      if (chunk < stride) {                      // wave-uniform: the chunk is inside the plane or beyond it
        if (lane < kWave / 2) {
#pragma unroll
          for (int c = 0; c < NC; ++c)
            __builtin_amdgcn_global_load_lds((global_void *)(base + (int64_t)c * stride + chunk + 2 * lane),
                                             (lds_void *)(uintptr_t)(uint32_t)(uintptr_t)(lds + ((k - REGK) * NC + c) * kWave),
                                             16, 0, 0);
        }
      } else {
#pragma unroll
        for (int c = 0; c < NC; ++c) lds[((k - REGK) * NC + c) * kWave + lane] = 0.0;
      }
    }
  }
}

// The same slots filled from the reference's AoS arrays (what pack_kernel + load_resident produce,
// value for value: the symmetric part of the column-major 3x3, zeros in the padding).
template <int NC, int CPL, int REGK>
__device__ __forceinline__ void load_resident_aos(const double *__restrict__ b1, const double *__restrict__ b2,
                                                  const double *__restrict__ cv, const double *__restrict__ ch,
                                                  int n, int first_corr, int lane, double (&d)[REGK][NC],
                                                  double *lds /* [CPL-REGK][NC][64] */) {
  auto put = [&](auto kc, int c, double v) {
    constexpr int k = decltype(kc)::value;
    if constexpr (k < REGK) d[k][c] = v;
    else lds[((k - REGK) * NC + c) * kWave + lane] = v;
  };
  auto load_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int idx = first_corr + slot_corr<CPL, REGK>(k, lane);
    const bool in = idx < n;
    const int64_t j = in ? idx : 0;  // a valid address for the masked lanes
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      put(kc, c, in ? b1[3 * j + c] : 0.0);
      put(kc, 3 + c, in ? b2[3 * j + c] : 0.0);
    }
    if constexpr (NC >= 12) {
      const double *C = cv + 9 * j;
      put(kc, 6, in ? C[0] : 0.0);
      put(kc, 7, in ? 0.5 * (C[1] + C[3]) : 0.0);
      put(kc, 8, in ? 0.5 * (C[2] + C[6]) : 0.0);
      put(kc, 9, in ? C[4] : 0.0);
      put(kc, 10, in ? 0.5 * (C[5] + C[7]) : 0.0);
      put(kc, 11, in ? C[8] : 0.0);
    }
    if constexpr (NC >= 18) {
      const double *C = ch + 9 * j;
      put(kc, 12, in ? C[0] : 0.0);
      put(kc, 13, in ? 0.5 * (C[1] + C[3]) : 0.0);
      put(kc, 14, in ? 0.5 * (C[2] + C[6]) : 0.0);
      put(kc, 15, in ? C[4] : 0.0);
      put(kc, 16, in ? 0.5 * (C[5] + C[7]) : 0.0);
      put(kc, 17, in ? C[8] : 0.0);
    }
  };
  load_slot(std::integral_constant<int, 0>{});
  if constexpr (CPL >= 2) load_slot(std::integral_constant<int, 1>{});
  if constexpr (CPL >= 4) {
    load_slot(std::integral_constant<int, 2>{});
    load_slot(std::integral_constant<int, 3>{});
  }
  if constexpr (CPL >= 8) {
    load_slot(std::integral_constant<int, 4>{});
    load_slot(std::integral_constant<int, 5>{});
    load_slot(std::integral_constant<int, 6>{});
    load_slot(std::integral_constant<int, 7>{});
  }
}

// One fused pass of a wavefront over its resident correspondences: acc = this lane's partial
// sums of r^2, J'r and J'J at the pose in U.
// nslots = how many of the wavefront's slots hold at least one correspondence (wave-uniform; slot k
// is empty for every lane once the pair's count is below the slot's first correspondence).
template <int MODE, int REGK, int LDSK>
__device__ __forceinline__ void pass_resident(const double (&d)[REGK][num_components(MODE)],
                                              const double *lds /* [LDSK][NC][64] */, int nslots,
                                              int lane, const PassUniforms &U, double reg,
                                              double (&acc)[kNumAcc]) {
  constexpr int NC = num_components(MODE);
  // The first two slots are always evaluated (padding contributes exact zeros, see eval_corr).  The
  // rest of the register slots, and each LDS slot, are skipped by a wave-uniform branch when no
  // lane holds a correspondence there -- the tail of a ragged pair -- so a short pair does
  // not pay for what the geometry could hold.  (Measured: < 0.5 % on full-size pairs,
  // +8 % solves/s on pairs of 513..600 correspondences.)
  auto eval_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double r, J[5];
    eval_corr<MODE>(d[k], U, reg, r, J);
    accumulate(r, J, acc);
  };
  eval_slot(std::integral_constant<int, 0>{});
  if constexpr (REGK > 1) eval_slot(std::integral_constant<int, 1>{});
  if constexpr (REGK > 2) {
    if (nslots > 2) {
      eval_slot(std::integral_constant<int, 2>{});
      if constexpr (REGK > 3) eval_slot(std::integral_constant<int, 3>{});
      if constexpr (REGK > 4) eval_slot(std::integral_constant<int, 4>{});
      if constexpr (REGK > 5) eval_slot(std::integral_constant<int, 5>{});
      if constexpr (REGK > 6) eval_slot(std::integral_constant<int, 6>{});
      if constexpr (REGK > 7) eval_slot(std::integral_constant<int, 7>{});
    }
  }
  // unrolled: with the max-ILP scheduler the slots' loads are issued ahead of their use without
  // blowing the register budget (+0.5 %; the default scheduler hoisted all of them and spilled)
#pragma unroll
  for (int k = 0; k < LDSK; ++k) {
    if (REGK + k >= nslots) continue;
    double e[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) e[c] = lds[(k * NC + c) * kWave + lane];
    double r, J[5];
    eval_corr<MODE>(e, U, reg, r, J);
    accumulate(r, J, acc);
  }
}

// The pass at the iteration cap: the candidate's cost decides accept / reject of the last step and
// the solve ends either way (Ceres tests max_num_iterations before it looks at the new gradient), so
// residual and propagated variance are all that is needed -- 40 of the full pass's 91 instructions
// per correspondence.  acc0 = sum r^2; z = 0, or NaN when a Jacobian entry would not be finite.
template <int MODE, int REGK, int LDSK>
__device__ __forceinline__ void pass_cost_resident(const double (&d)[REGK][num_components(MODE)],
                                                   const double *lds /* [LDSK][NC][64] */, int nslots,
                                                   int lane, const PassUniforms &U, double reg,
                                                   double &acc0, double &z) {
  constexpr int NC = num_components(MODE);
  auto eval_slot = [&](auto kc) {
    constexpr int k = decltype(kc)::value;
    double r, kk;
    eval_cost<MODE>(d[k], U, reg, r, kk);
    acc0 = __builtin_fma(r, r, acc0);
    z = __builtin_fma(kk, 0.0, z);
  };
  eval_slot(std::integral_constant<int, 0>{});
  if constexpr (REGK > 1) eval_slot(std::integral_constant<int, 1>{});
  if constexpr (REGK > 2) {
    if (nslots > 2) {
      eval_slot(std::integral_constant<int, 2>{});
      if constexpr (REGK > 3) eval_slot(std::integral_constant<int, 3>{});
      if constexpr (REGK > 4) eval_slot(std::integral_constant<int, 4>{});
      if constexpr (REGK > 5) eval_slot(std::integral_constant<int, 5>{});
      if constexpr (REGK > 6) eval_slot(std::integral_constant<int, 6>{});
      if constexpr (REGK > 7) eval_slot(std::integral_constant<int, 7>{});
    }
  }
#pragma unroll
  for (int k = 0; k < LDSK; ++k) {
    if (REGK + k >= nslots) continue;
    double e[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) e[c] = lds[(k * NC + c) * kWave + lane];
    double r, kk;
    eval_cost<MODE>(e, U, reg, r, kk);
    acc0 = __builtin_fma(r, r, acc0);
    z = __builtin_fma(kk, 0.0, z);
  }
}

// pose -> the correspondence-independent quantities of a pass, as plain doubles
__device__ __forceinline__ void pose_uniforms_sc(double st, double ct, double sp, double cp, const double (&q)[4],
                                                 double *u) {
  double R[9];
  rot_from_quat(q, R);
#pragma unroll
  for (int i = 0; i < 9; ++i) u[i] = R[i];
  u[9] = st * cp;   u[10] = st * sp;  u[11] = ct;
  u[12] = ct * cp;  u[13] = ct * sp;  u[14] = -st;
  // (dt/dphi = (-u[10], u[9], 0): the pass derives it from t)
}
__device__ __forceinline__ void pose_uniforms(double theta, double phi, const double (&q)[4], double *u) {
  double st, ct, sp, cp;
  sincos_bounded(theta, st, ct);
  sincos_bounded(phi, sp, cp);
  pose_uniforms_sc(st, ct, sp, cp, q, u);
}

// PNECCeres::Result(): pnec_ceres.cc:201-207
__device__ __forceinline__ void write_result(const SolveArgs &a, int64_t s, const double *slab, int iteration,
                                             int term) {
  const double q0 = slab[kQ + 0], q1 = slab[kQ + 1], q2 = slab[kQ + 2], q3 = slab[kQ + 3];
  const double qn = fast_rsqrt(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
  if (a.out_q) {
    a.out_q[4 * s + 0] = q0 * qn;
    a.out_q[4 * s + 1] = q1 * qn;
    a.out_q[4 * s + 2] = q2 * qn;
    a.out_q[4 * s + 3] = q3 * qn;
  }
  if (a.out_t) {
    // t(theta, phi) = (sin th cos ph, sin th sin ph, cos th) of the current point: the very products its pass
    // evaluated with (pose_uniforms_sc), parked by lm_advance when the point was accepted
    a.out_t[3 * s + 0] = slab[kTcur + 0];
    a.out_t[3 * s + 1] = slab[kTcur + 1];
    a.out_t[3 * s + 2] = slab[kTcur + 2];
  }
  if (a.out_cost) a.out_cost[s] = slab[kCost];
  if (a.out_iterations) a.out_iterations[s] = iteration;
  if (a.out_status) a.out_status[s] = term;
}

// Advance ONE solve (called by the four lanes of a quad, which all do the same): consume the sums of the pass at the candidate, run
// Ceres' accept/reject + trust-region logic (TrustRegionMinimizer + LevenbergMarquardtStrategy,
// SURVEY.md Appendix B), and either publish the next candidate (slab + pass uniforms) or
// terminate.  Returns the termination code, or -1 while the solve goes on.
//
// This is a latency chain (~500 dependent FP64 instructions) executed by one lane while the other
// wavefront of the SIMD runs its pass, so it is written to keep LDS round trips off the chain:
// everything is loaded in one batch up front, values are forwarded in registers (an accepted
// point's J'J / J'r are the pass's sums themselves), and the stores trail.
// COST_FIRST (round 4): after a REJECTED step the next candidate's pass is cost-only (ist[kILast] = 2), the way Ceres
// itself evaluates a candidate -- residuals first, the Jacobian only once the step is accepted.  The fused pass
// speculates on acceptance (cost, J'J and J'r in one go: 91 instructions per correspondence); for a solve that sits at
// its noise floor the speculation mostly loses: on the benchmark's batch (ten iterations whatever happens; the solves
// converge after three or four) 42 % of all steps are rejected, and 83 % of the steps that follow a rejected one
// (counted on the CPU checker: DESIGN.md 6).  The cost-only pass is 43 instructions per correspondence and one sum
// through the reduction tree instead of 21; when such a step IS accepted the same candidate is evaluated once more in
// full and this function runs again on the complete sums -- same cost bits (the cost-only pass reduces through the same
// tree), same decision, same everything downstream: the results are bit for bit those of the always-speculating kernel.
// With Ceres-default termination there is next to nothing to gain or lose (18 rejected steps in 10 046).
template <bool COST_FIRST>
__device__ __forceinline__ int lm_advance(double *slab, int *ist, double *unif, const pnec_hip_options &o,
                                          double inv_max_radius, double inv_min_radius) {
  int iteration = ist[kIIter], reuse_diagonal = ist[kIReuseDiag];
  int num_invalid = ist[kINumInvalid], step_ok = ist[kIStepOk];
  const int first = ist[kIFirst];
  const int pass_kind = ist[kILast];     // 0: full pass; 1: cost-only at the iteration cap; 2: cost-only after a rejected step
  const bool last = pass_kind == 1;      // the pass was the cost-only one: sums 1..20 do not exist
  const bool cost_first = COST_FIRST && pass_kind == 2;
  int park = ist[kIPark];
  int term = -1;

  // ---- one batch of loads (the candidate's sums: the table the current point does not own)
  double S[kNumAcc];
  const double *cand_sums = slab + kSums + (park ^ 1) * kSumSlots;
#pragma unroll
  for (int j = 0; j < kNumAcc; ++j) S[j] = cand_sums[sum_slot(j)];
  const bool rest_ok = slab[kSumsFinite] != 0.0;
  const double qc0 = slab[kQc + 0], qc1 = slab[kQc + 1], qc2 = slab[kQc + 2], qc3 = slab[kQc + 3];
  const double thc0 = slab[kThetaC], phc0 = slab[kPhiC];
  const double cost = slab[kCost], model = slab[kModel], xnorm = slab[kXNorm];
  // The trust region is carried as 1 / radius: LevenbergMarquardtStrategy only ever divides by the
  // radius (D / radius), the update radius / max(1/3, 1 - (2 rho - 1)^3) is a multiplication of the
  // inverse, and the two bounds are comparisons against 1 / max_radius and 1 / min_radius -- no
  // reciprocal on the chain.  dec = Ceres' decrease_factor (2, 4, 8, ...: exact).
  double inv_radius = slab[kInvRadius], dec = slab[kDec], gmax = slab[kGmax];
  double x[6];

  double cost_c = 0.5 * S[0];
  const bool cost_ok = finite_d(cost_c);
  bool accept = false;
  double rho = 0.0;
  if (first) {
    if (!(cost_ok && rest_ok)) {
      slab[kQ + 0] = qc0; slab[kQ + 1] = qc1; slab[kQ + 2] = qc2; slab[kQ + 3] = qc3;
      slab[kTheta] = thc0;
      slab[kPhi] = phc0;
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
      slab[kCost] = cost_c;
      term = PNEC_HIP_TERM_BAD_INITIAL;
    } else {
      // jacobi_scaling: s = 1 / (1 + sqrt(diag(J'J))) on the Ceres-tangent Jacobian (rotation
      // columns = 2 x the omega columns accumulated by the pass); frozen after iteration zero.
      // Parked: (f s)^2 and its inverse, f = 2 for the rotation columns.
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const double f = (i >= 2) ? 2.0 : 1.0, inv_f = (i >= 2) ? 0.5 : 1.0;
        const double hii = S[6 + tri(i, i)] * (f * f);
        const double a = (o.jacobi_scaling ? 1.0 + fast_sqrt(hii) : 1.0) * inv_f;  // 1 / (f s)
        slab[kInvScaleSq + i] = a * a;
        slab[kScaleSq + i] = fast_rcp(a * a);
      }
      accept = true;
    }
  } else {
    if (!cost_ok) cost_c = 1.7976931348623157e308;
    if (o.check_convergence) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = slab[kQ + k];
      x[4] = slab[kTheta];
      x[5] = slab[kPhi];
      double dn = (x[4] - thc0) * (x[4] - thc0) + (x[5] - phc0) * (x[5] - phc0);
      dn = __builtin_fma(x[0] - qc0, x[0] - qc0, dn);
      dn = __builtin_fma(x[1] - qc1, x[1] - qc1, dn);
      dn = __builtin_fma(x[2] - qc2, x[2] - qc2, dn);
      dn = __builtin_fma(x[3] - qc3, x[3] - qc3, dn);
      const double step_norm = fast_sqrt(dn);
      if (step_norm <= o.parameter_tolerance * (xnorm + o.parameter_tolerance))
        term = PNEC_HIP_TERM_PARAMETER_TOL;
      else if (fabs(cost - cost_c) <= o.function_tolerance * cost)
        term = PNEC_HIP_TERM_FUNCTION_TOL;
    }
    if (term < 0) {
      rho = (cost - cost_c) * fast_rcp(model);
      accept = rho > o.min_relative_decrease;
      if (accept && !rest_ok) term = PNEC_HIP_TERM_BAD_INITIAL;  // finite cost, non-finite Jacobian: Ceres fails here
    }
  }

  if (term < 0 && cost_first && accept) {
    // accepted on its cost alone: its normal equations are needed now -- the same candidate once more, in full.
    // Nothing has been changed up to here (no counter, no radius, no table): the next call starts from the same state.
    ist[kILast] = 0;
    return -1;
  }
  if (term < 0 && last) {
    // at the iteration cap the solve ends here whatever the verdict on the step (Ceres checks
    // max_num_iterations before anything that would read the new Jacobian)
    if (accept) {
      slab[kQ + 0] = qc0; slab[kQ + 1] = qc1; slab[kQ + 2] = qc2; slab[kQ + 3] = qc3;
      slab[kTheta] = thc0;
      slab[kPhi] = phc0;
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
      slab[kCost] = cost_c;
    }
    term = PNEC_HIP_TERM_MAX_ITERATIONS;
  }
  if (term < 0) {
    double H[15], g[5], diag[5];
    if (accept) {
      // x <- candidate; its normal equations are the sums of the pass just made
      x[0] = qc0; x[1] = qc1; x[2] = qc2; x[3] = qc3; x[4] = thc0; x[5] = phc0;
#pragma unroll
      for (int i = 0; i < 5; ++i) g[i] = S[1 + i];
#pragma unroll
      for (int i = 0; i < 15; ++i) H[i] = S[6 + i];
      if (o.check_convergence) {  // only the gradient-tolerance test reads it
        gmax = 0.0;
#pragma unroll
        for (int i = 0; i < 5; ++i) gmax = fmax(gmax, fabs(g[i]) * ((i >= 2) ? 2.0 : 1.0));
      }
      if (first) {
        inv_radius = fast_rcp(o.initial_trust_region_radius);
      } else {
        const double c1 = 2.0 * rho - 1.0;
        inv_radius = fmax(inv_max_radius, inv_radius * fmax(1.0 / 3.0, 1.0 - c1 * c1 * c1));
      }
      dec = 2.0;
      park ^= 1;  // the candidate's table is the current point's from now on (nothing is copied)
      step_ok = 1;
      reuse_diagonal = 0;
      // park (stores only; nothing below reads them back)
#pragma unroll
      for (int k = 0; k < 4; ++k) slab[kQ + k] = x[k];
      slab[kTheta] = x[4];
      slab[kPhi] = x[5];
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
      slab[kCost] = cost_c;
      if (o.check_convergence) {  // |x| is only read by the parameter-tolerance test
        slab[kXNorm] = fast_sqrt(x[4] * x[4] + x[5] * x[5] + x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
        slab[kGmax] = gmax;
      }
    } else {
      // rejected: back to the parked point (its normal equations are still in its table), smaller region
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = slab[kQ + k];
      x[4] = slab[kTheta];
      x[5] = slab[kPhi];
      const double *cur_sums = slab + kSums + park * kSumSlots;
#pragma unroll
      for (int i = 0; i < 5; ++i) g[i] = cur_sums[sum_slot(1 + i)];
#pragma unroll
      for (int i = 0; i < 15; ++i) H[i] = cur_sums[sum_slot(6 + i)];
      inv_radius = inv_radius * dec;
      dec = 2.0 * dec;
      reuse_diagonal = 1;
    }
    if (reuse_diagonal) {
#pragma unroll
      for (int i = 0; i < 5; ++i) diag[i] = slab[kDiag + i];
    }

    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue + the next trust-region step
    for (bool retry = false;; retry = true) {
      if (iteration >= o.max_num_iterations) { term = PNEC_HIP_TERM_MAX_ITERATIONS; break; }
      if (retry) {
        // (rare) the previous attempt consumed H in place: fetch it from the current point's table again
        const double *cur_sums = slab + kSums + park * kSumSlots;
#pragma unroll
        for (int i = 0; i < 15; ++i) H[i] = cur_sums[sum_slot(6 + i)];
      }
      if (o.check_convergence && step_ok && gmax <= o.gradient_tolerance) {
        term = PNEC_HIP_TERM_GRADIENT_TOL; break;
      }
      if (inv_radius > inv_min_radius) { term = PNEC_HIP_TERM_MIN_RADIUS; break; }  // radius < min_radius
      ++iteration;
      step_ok = 0;

      // LevenbergMarquardtStrategy::ComputeStep.  Ceres solves (S H S + D/radius) y = -S g with the
      // Jacobi scaling S frozen at iteration zero, D = clamp(diag(S H S)), and steps by S y.  With
      // p = S y that is (H + S^-1 D S^-1 / radius) p = -g: the scaling only enters through the
      // diagonal D' = clamp(s_i^2 H_ii) / s_i^2, and p is the parameter step itself
      // (theta, phi, omega; the quaternion's half-angle delta = omega / 2).
      if (!reuse_diagonal) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          diag[i] = fmin(fmax(H[tri(i, i)] * slab[kScaleSq + i], o.min_lm_diagonal), o.max_lm_diagonal) *
                    slab[kInvScaleSq + i];
          slab[kDiag + i] = diag[i];
        }
      }
      double dr[5], y[5], step[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        dr[i] = diag[i] * inv_radius;
        H[tri(i, i)] += dr[i];
      }
      bool valid = chol_solve5(H, g, y);
      // model cost change -(Jp)'(r + Jp/2) = -(g'p + p'Hp/2); with (H + D'/radius) p = -g this is
      // (-g'p + p'(D'/radius)p) / 2 -- two non-negative terms, no cancellation
      double sg = 0.0, sd = 0.0;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        step[i] = -y[i];
        sg = __builtin_fma(step[i], g[i], sg);
        sd = __builtin_fma(dr[i] * step[i], step[i], sd);
      }
      const double model_change = 0.5 * (sd - sg);
      valid = valid && (model_change > 0.0);
      if (!valid) {
        if (++num_invalid >= o.max_num_consecutive_invalid_steps) { term = PNEC_HIP_TERM_INVALID_STEPS; break; }
        // [EXT, recalled] TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid():
        // radius *= 0.5, reuse_diagonal = true -- NOT the rejected-step rule: decrease_factor stays as it is
        inv_radius = 2.0 * inv_radius;
        reuse_diagonal = 1;
        continue;
      }
      num_invalid = 0;

      // candidate = Plus(x, p): EigenQuaternionManifold::Plus on q with delta = omega / 2
      const double dx = 0.5 * step[2], dy = 0.5 * step[3], dz = 0.5 * step[4];
      const double thc = x[4] + step[0];
      const double phc = x[5] + step[1];
      const double nd2 = dx * dx + dy * dy + dz * dz;
      const double ind = nd2 > 0.0 ? fast_rsqrt(nd2) : 0.0, nd = nd2 * ind;
      // the three sine/cosine pairs of the step (|delta|, theta, phi) in ONE evaluation: the caller
      // runs this function on the four lanes of a quad, identical in all of them up to here; lane
      // 0 / 1 / 2 takes its own angle and the results are exchanged inside the quad
      const int role = (int)(threadIdx.x & 3);
      const double angle = role == 0 ? nd : (role == 1 ? thc : phc);
      double sa, ca;
      sincos_bounded_pinned(angle, sa, ca);
      const double sn = quad_broadcast<0>(sa), aw = quad_broadcast<0>(ca);
      const double st = quad_broadcast<1>(sa), ct = quad_broadcast<1>(ca);
      const double sp = quad_broadcast<2>(sa), cp = quad_broadcast<2>(ca);
      const double sbd = sn * ind;  // sin|delta| / |delta| (0 for a zero step: qc = x)
      const double ax = sbd * dx, ay = sbd * dy, az = sbd * dz;
      double qc[4];
      qc[0] = aw * x[0] + ax * x[3] + ay * x[2] - az * x[1];
      qc[1] = aw * x[1] - ax * x[2] + ay * x[3] + az * x[0];
      qc[2] = aw * x[2] + ax * x[1] - ay * x[0] + az * x[3];
      qc[3] = aw * x[3] - ax * x[0] - ay * x[1] - az * x[2];
      pose_uniforms_sc(st, ct, sp, cp, qc, unif);
#pragma unroll
      for (int k = 0; k < 4; ++k) slab[kQc + k] = qc[k];
      slab[kThetaC] = thc;
      slab[kPhiC] = phc;
      slab[kModel] = model_change;
      // the next pass: cost-only at the cap; cost-only too when the step just decided was rejected, or when the model
      // promises less than the cost can resolve (a decrease below ~4e-15 of the cost is inside the rounding of a sum of
      // 512 squares: on the benchmark's batch 65 % of such steps end rejected even right after an accepted one, and
      // every step that promises more than 1e-14 is accepted -- histogram from the CPU checker)
      const double cost_now = accept ? cost_c : cost;
      ist[kILast] = iteration >= o.max_num_iterations ? 1 : ((COST_FIRST && (!accept || model_change <= 4.0e-15 * cost_now)) ? 2 : 0);
      break;
    }
    slab[kInvRadius] = inv_radius;
    slab[kDec] = dec;
  }
  ist[kIPark] = park;

  ist[kIIter] = iteration;
  ist[kIFirst] = (term < 0 || !first) ? 0 : 1;
  ist[kIReuseDiag] = reuse_diagonal;
  ist[kINumInvalid] = num_invalid;
  ist[kIStepOk] = step_ok;
  return term;
}

// ---- the same step, ACROSS THE LANES OF A ROW: an A/B build (-DPNEC_ADVANCE_ROWS), NOT the default.  Measured on the
// benchmark (same box, same call): SQ_INSTS_VALU 14 202 -> 13 936 per solve (-1.9 %), 33.8 -> 33.1 M solves/s (-2 %).
// The kernel sits where two limits meet -- the VALU issue slots of two wavefronts per SIMD and the latency of this
// chain against the other wavefront's pass -- and this form buys its fewer slots with a longer chain (the gather is
// two dependent LDS round trips instead of one, every fused broadcast waits out the DPP hazard).  Kept because it is
// the measured answer to "run the 5x5 solve across lanes" and the home of the DP-ALU DPP primitives.
// lm_advance runs one solve's step identically in four lanes: a wavefront instruction costs its four issue
// clocks whether four or sixty-four lanes are live, so everything that is a 5-vector or a 5x5 matrix is paid for
// five (or fifteen) times over.  Here the WHOLE wavefront executes the step (EXEC full: the 64-bit DPP broadcasts
// need their source lanes active) and lane i (i = 0..4) of every 16-lane row owns component i: row i of the normal
// equations, g_i, the LM diagonal, the Jacobi scale, the step p_i.  Cross-lane traffic is the DP ALU's
// row_newbcast fused into v_fmac_f64 (pnec_device.hpp: fmac_row / fnmac_row / bcast_row), i.e. free of its own
// instruction: the 5x5 system is a Gauss-Jordan elimination across lanes (gj_solve5_rows, ~70 slots against ~125),
// the clamped diagonal, the scaled step and the model decrease are one lane-parallel expression each, and the six
// sines / cosines of the step reach the quaternion and the pose uniforms through one 64-bit broadcast each instead
// of two 32-bit DPP moves.  Scalars of the solve (cost, radius, iteration counters, the quaternion) stay replicated in
// every lane exactly as before; the control flow is the same code, statement for statement.
//
// Lane i finds its row in the 24-slot tables of sums through `gidx` (LDS, 8 ints per lane, written once per
// kernel): byte offsets of H(i,0..4) and g_i inside a table (lanes >= 5: offset 0, harmless).  The four rows of
// the wavefront do the same thing on the same data; stores are made by lane 0 (or lanes 0..4) only.
constexpr int kGatherInts = 8;  // per lane: H(i,0) .. H(i,4), g_i, H(i,i), pad   (byte offsets into a table of sums)
__device__ __forceinline__ void gather_index_init(int *gidx, int lane) {
  if (lane < 16) {
    int v[kGatherInts];
#pragma unroll
    for (int k = 0; k < kGatherInts; ++k) v[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (lane == i) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = 8 * sum_slot(6 + tri(i < k ? i : k, i < k ? k : i));
        v[5] = 8 * sum_slot(1 + i);
        v[6] = 8 * sum_slot(6 + tri(i, i));
      }
#pragma unroll
    for (int k = 0; k < kGatherInts; ++k) gidx[lane * kGatherInts + k] = v[k];
  }
}

__device__ __forceinline__ int lm_advance_rows(double *slab, int *ist, double *unif, const int *gidx,
                                               const pnec_hip_options &o, double inv_max_radius,
                                               double inv_min_radius, int lane) {
  const int li = lane & 15;
  const bool own = li < 5;          // this lane owns a component
  const bool writer = lane == 0;    // the one lane that stores scalars
  const bool cwriter = lane < 5;    // the lanes that store 5-vectors
  int iteration = ist[kIIter], reuse_diagonal = ist[kIReuseDiag];
  int num_invalid = ist[kINumInvalid], step_ok = ist[kIStepOk];
  const int first = ist[kIFirst];
  const bool last = ist[kILast] != 0;  // the pass was the cost-only one: sums 1..20 do not exist
  int park = ist[kIPark];
  int term = -1;

  // ---- loads: the scalars (broadcast reads) and this lane's gather offsets
  const char *cand_tab = reinterpret_cast<const char *>(slab + kSums + (park ^ 1) * kSumSlots);
  const double S0 = *reinterpret_cast<const double *>(cand_tab + 8 * sum_slot(0));
  const bool rest_ok = slab[kSumsFinite] != 0.0;
  const double qc0 = slab[kQc + 0], qc1 = slab[kQc + 1], qc2 = slab[kQc + 2], qc3 = slab[kQc + 3];
  const double thc0 = slab[kThetaC], phc0 = slab[kPhiC];
  const double cost = slab[kCost], model = slab[kModel], xnorm = slab[kXNorm];
  double inv_radius = slab[kInvRadius], dec = slab[kDec], gmax = slab[kGmax];
  int go[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) go[k] = gidx[li * kGatherInts + k];
  auto row_of = [&](const char *tab, double (&A)[5], double &gi) {
#pragma unroll
    for (int k = 0; k < 5; ++k) A[k] = *reinterpret_cast<const double *>(tab + go[k]);
    gi = *reinterpret_cast<const double *>(tab + go[5]);
  };
  // this lane's own diagonal entry H(i,i) is column i of its row: its own gather offset, not a register select
  auto diag_of = [&](const char *tab) { return *reinterpret_cast<const double *>(tab + go[6]); };
  const double fcol = li >= 2 ? 2.0 : 1.0;  // the rotation columns of Ceres' tangent Jacobian are 2 x the pass's
  double x[6];

  double cost_c = 0.5 * S0;
  const bool cost_ok = finite_d(cost_c);
  bool accept = false;
  double rho = 0.0;
  if (first) {
    if (!(cost_ok && rest_ok)) {
      if (writer) {
        slab[kQ + 0] = qc0; slab[kQ + 1] = qc1; slab[kQ + 2] = qc2; slab[kQ + 3] = qc3;
        slab[kTheta] = thc0;
        slab[kPhi] = phc0;
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
        slab[kCost] = cost_c;
      }
      term = PNEC_HIP_TERM_BAD_INITIAL;
    } else {
      // jacobi_scaling: s = 1 / (1 + sqrt(diag(J'J))) on the Ceres-tangent Jacobian; frozen after iteration zero.
      // Parked: (f s)^2 and its inverse, f = 2 for the rotation columns.  One component per lane.
      const double hii = diag_of(cand_tab) * (fcol * fcol);
      const double a = (o.jacobi_scaling ? 1.0 + fast_sqrt(hii) : 1.0) * (li >= 2 ? 0.5 : 1.0);  // 1 / (f s)
      if (cwriter) {
        slab[kInvScaleSq + li] = a * a;
        slab[kScaleSq + li] = fast_rcp(a * a);
      }
      accept = true;
    }
  } else {
    if (!cost_ok) cost_c = 1.7976931348623157e308;
    if (o.check_convergence) {
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = slab[kQ + k];
      x[4] = slab[kTheta];
      x[5] = slab[kPhi];
      double dn = (x[4] - thc0) * (x[4] - thc0) + (x[5] - phc0) * (x[5] - phc0);
      dn = __builtin_fma(x[0] - qc0, x[0] - qc0, dn);
      dn = __builtin_fma(x[1] - qc1, x[1] - qc1, dn);
      dn = __builtin_fma(x[2] - qc2, x[2] - qc2, dn);
      dn = __builtin_fma(x[3] - qc3, x[3] - qc3, dn);
      const double step_norm = fast_sqrt(dn);
      if (step_norm <= o.parameter_tolerance * (xnorm + o.parameter_tolerance))
        term = PNEC_HIP_TERM_PARAMETER_TOL;
      else if (fabs(cost - cost_c) <= o.function_tolerance * cost)
        term = PNEC_HIP_TERM_FUNCTION_TOL;
    }
    if (term < 0) {
      rho = (cost - cost_c) * fast_rcp(model);
      accept = rho > o.min_relative_decrease;
      if (accept && !rest_ok) term = PNEC_HIP_TERM_BAD_INITIAL;  // finite cost, non-finite Jacobian: Ceres fails here
    }
  }

  if (term < 0 && last) {
    // at the iteration cap the solve ends here whatever the verdict on the step
    if (accept && writer) {
      slab[kQ + 0] = qc0; slab[kQ + 1] = qc1; slab[kQ + 2] = qc2; slab[kQ + 3] = qc3;
      slab[kTheta] = thc0;
      slab[kPhi] = phc0;
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
      slab[kCost] = cost_c;
    }
    term = PNEC_HIP_TERM_MAX_ITERATIONS;
  }
  if (term < 0) {
    double Hr[5], gi, diag = 0.0;   // this lane's row of J'J, its g_i, its LM diagonal entry
    const char *cur_tab;            // the table of the point the next step starts from
    if (accept) {
      // x <- candidate; its normal equations are the sums of the pass just made
      x[0] = qc0; x[1] = qc1; x[2] = qc2; x[3] = qc3; x[4] = thc0; x[5] = phc0;
      cur_tab = cand_tab;
      row_of(cur_tab, Hr, gi);
      if (o.check_convergence) {  // only the gradient-tolerance test reads it: max_i |g_i| f_i over the five lanes
        const double ga = fabs(gi) * fcol;
        gmax = fmax(fmax(fmax(bcast_row<0>(ga), bcast_row<1>(ga)), fmax(bcast_row<2>(ga), bcast_row<3>(ga))), bcast_row<4>(ga));
      }
      if (first) {
        inv_radius = fast_rcp(o.initial_trust_region_radius);
      } else {
        const double c1 = 2.0 * rho - 1.0;
        inv_radius = fmax(inv_max_radius, inv_radius * fmax(1.0 / 3.0, 1.0 - c1 * c1 * c1));
      }
      dec = 2.0;
      park ^= 1;  // the candidate's table is the current point's from now on (nothing is copied)
      step_ok = 1;
      reuse_diagonal = 0;
      if (writer) {
#pragma unroll
        for (int k = 0; k < 4; ++k) slab[kQ + k] = x[k];
        slab[kTheta] = x[4];
        slab[kPhi] = x[5];
        slab[kTcur + 0] = unif[9]; slab[kTcur + 1] = unif[10]; slab[kTcur + 2] = unif[11];
        slab[kCost] = cost_c;
        if (o.check_convergence) {  // |x| is only read by the parameter-tolerance test
          slab[kXNorm] = fast_sqrt(x[4] * x[4] + x[5] * x[5] + x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
          slab[kGmax] = gmax;
        }
      }
    } else {
      // rejected: back to the parked point (its normal equations are still in its table), smaller region
#pragma unroll
      for (int k = 0; k < 4; ++k) x[k] = slab[kQ + k];
      x[4] = slab[kTheta];
      x[5] = slab[kPhi];
      cur_tab = reinterpret_cast<const char *>(slab + kSums + park * kSumSlots);
      row_of(cur_tab, Hr, gi);
      inv_radius = inv_radius * dec;
      dec = 2.0 * dec;
      reuse_diagonal = 1;
    }
    const double hii = diag_of(cur_tab);
    if (reuse_diagonal) diag = slab[kDiag + (own ? li : 0)];
    const double scale_sq = slab[kScaleSq + (own ? li : 0)], inv_scale_sq = slab[kInvScaleSq + (own ? li : 0)];
    const double xang = li == 0 ? x[4] : x[5];   // lane 0 steps theta, lane 1 phi

    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue + the next trust-region step
    for (;;) {
      if (iteration >= o.max_num_iterations) { term = PNEC_HIP_TERM_MAX_ITERATIONS; break; }
      if (o.check_convergence && step_ok && gmax <= o.gradient_tolerance) {
        term = PNEC_HIP_TERM_GRADIENT_TOL; break;
      }
      if (inv_radius > inv_min_radius) { term = PNEC_HIP_TERM_MIN_RADIUS; break; }  // radius < min_radius
      ++iteration;
      step_ok = 0;

      // LevenbergMarquardtStrategy::ComputeStep in parameter scale (see lm_advance): (H + D'/radius) p = -g,
      // D'_i = clamp(s_i^2 H_ii) / s_i^2 -- one component per lane
      if (!reuse_diagonal) {
        diag = fmin(fmax(hii * scale_sq, o.min_lm_diagonal), o.max_lm_diagonal) * inv_scale_sq;
        if (cwriter) slab[kDiag + li] = diag;
      }
      const double dr = diag * inv_radius;
      double A[5];   // the elimination works in place: a retry starts from the row again
#pragma unroll
      for (int k = 0; k < 5; ++k) A[k] = Hr[k];
      double p = -gi;
      bool valid = gj_solve5_rows(A, dr, p, li);   // p_i = step component i (parameter scale), lanes 0..4
      // model cost change (-g'p + p'(D'/radius)p) / 2: the lanes' terms summed through broadcasts; a non-finite
      // step component makes the sum NaN (0 x inf), and NaN > 0 is false
      double e = __builtin_fma(dr * p, p, -(p * gi));
      e = __builtin_fma(p, 0.0, e);
      double msum = 0.0;
      fmac_row<0>(msum, e, 0.5);
      fmac_row<1>(msum, e, 0.5);
      fmac_row<2>(msum, e, 0.5);
      fmac_row<3>(msum, e, 0.5);
      fmac_row<4>(msum, e, 0.5);
      const double model_change = msum;
      valid = valid && (model_change > 0.0);
      if (!valid) {
        if (++num_invalid >= o.max_num_consecutive_invalid_steps) { term = PNEC_HIP_TERM_INVALID_STEPS; break; }
        // [EXT, recalled] TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid():
        // radius *= 0.5, reuse_diagonal = true -- NOT the rejected-step rule: decrease_factor stays as it is
        inv_radius = 2.0 * inv_radius;
        reuse_diagonal = 1;
        continue;
      }
      num_invalid = 0;

      // candidate = Plus(x, p): theta + p_0 (lane 0), phi + p_1 (lane 1); EigenQuaternionManifold::Plus on q with
      // delta = (p_2, p_3, p_4) / 2 (lanes 2, 3, 4)
      const double hp = 0.5 * p;
      const double hp2 = hp * hp;
      double nd2 = 0.0;
      fmac_row<2>(nd2, hp2, 1.0);
      fmac_row<3>(nd2, hp2, 1.0);
      fmac_row<4>(nd2, hp2, 1.0);
      const double ind = nd2 > 0.0 ? fast_rsqrt(nd2) : 0.0, nd = nd2 * ind;
      // the three sine / cosine pairs of the step in ONE evaluation: lane 0 theta, lane 1 phi, lane 2 |delta|
      const double angle = li == 2 ? nd : xang + p;
      double sa, ca;
      sincos_bounded(angle, sa, ca);
      const double st = bcast_row<0>(sa), ct = bcast_row<0>(ca);
      const double sp = bcast_row<1>(sa), cp = bcast_row<1>(ca);
      const double aw = bcast_row<2>(ca);
      double sbd = 0.0;                 // sin|delta| / |delta| (0 for a zero step: qc = x)
      fmac_row<2>(sbd, sa, ind);
      double ax = 0.0, ay = 0.0, az = 0.0;
      fmac_row<2>(ax, hp, sbd);
      fmac_row<3>(ay, hp, sbd);
      fmac_row<4>(az, hp, sbd);
      double qc[4];
      qc[0] = aw * x[0] + ax * x[3] + ay * x[2] - az * x[1];
      qc[1] = aw * x[1] - ax * x[2] + ay * x[3] + az * x[0];
      qc[2] = aw * x[2] + ax * x[1] - ay * x[0] + az * x[3];
      qc[3] = aw * x[3] - ax * x[0] - ay * x[1] - az * x[2];
      const double thc = bcast_row<0>(angle), phc = bcast_row<1>(angle);
      if (writer) {
        pose_uniforms_sc(st, ct, sp, cp, qc, unif);
#pragma unroll
        for (int k = 0; k < 4; ++k) slab[kQc + k] = qc[k];
        slab[kThetaC] = thc;
        slab[kPhiC] = phc;
        slab[kModel] = model_change;
        ist[kILast] = iteration >= o.max_num_iterations ? 1 : 0;
      }
      break;
    }
    if (writer) {
      slab[kInvRadius] = inv_radius;
      slab[kDec] = dec;
    }
  }
  if (writer) {
    ist[kIPark] = park;
    ist[kIIter] = iteration;
    ist[kIFirst] = (term < 0 || !first) ? 0 : 1;
    ist[kIReuseDiag] = reuse_diagonal;
    ist[kINumInvalid] = num_invalid;
    ist[kIStepOk] = step_ok;
  }
  return term;
}

// ---- PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL: the reference's own differentiation, for verification --------------------
// ceres::NumericDiffCostFunction<Functor, CENTRAL, 1, 1, 1, 4> (pnec_ceres.cc:84-97) [EXT, SURVEY Appendix B]: per ambient
// parameter x_j of (theta, phi, qx, qy, qz, qw): h = max(sqrt(eps), 1e-6 |x_j|), J_j = (r(x + h e_j) - r(x - h e_j)) / 2h;
// the quaternion's components are perturbed one by one WITHOUT renormalisation (the functor's toRotationMatrix is
// Eigen's un-normalised formula, rot_from_quat), and EigenQuaternionManifold::PlusJacobian (4x3) maps the 1x4 block to
// the tangent space.  Twelve perturbed poses per pass: lane j < 12 of the first wavefront makes pose j = 2 * parameter +
// (0: plus, 1: minus) and leaves its R | t (12 doubles) and 1 / (2 h) in LDS; every lane then evaluates its
// correspondences' residual at the centre and at the twelve poses.
constexpr int kNumPoses = 12;
__device__ __forceinline__ void numeric_poses(const double *slab, int lane, double (*nu)[12], double *inv2h) {
  if (lane < kNumPoses) {
    const int prm = lane >> 1;
    double x[6] = {slab[kThetaC], slab[kPhiC], slab[kQc + 0], slab[kQc + 1], slab[kQc + 2], slab[kQc + 3]};
    const double xv = prm == 0 ? x[0] : (prm == 1 ? x[1] : (prm == 2 ? x[2] : (prm == 3 ? x[3] : (prm == 4 ? x[4] : x[5]))));
    const double h = fmax(1.4901161193847656e-08, fabs(xv) * 1e-6);
    const double xp = (lane & 1) ? xv - h : xv + h;
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = (k == prm) ? xp : x[k];
    double u[kUnif];
    const double q[4] = {x[2], x[3], x[4], x[5]};
    pose_uniforms(x[0], x[1], q, u);
#pragma unroll
    for (int i = 0; i < 12; ++i) nu[lane][i] = u[i];
    if ((lane & 1) == 0) inv2h[prm] = 1.0 / h / 2.0;
  }
}
template <int MODE>
__device__ __forceinline__ void eval_corr_numeric(const double (&e)[num_components(MODE)], const PassUniforms &U,
                                                  const double (*nu)[12], const double *inv2h, const double *qc,
                                                  double reg, double &r, double (&J)[5]) {
  double kk;
  eval_cost<MODE>(e, U, reg, r, kk);
  double Ja[6];
#pragma unroll 1
  for (int prm = 0; prm < 6; ++prm) {
    double rr[2];
#pragma unroll
    for (int sgn = 0; sgn < 2; ++sgn) {
      PassUniforms V;
      const double *u = nu[2 * prm + sgn];
#pragma unroll
      for (int i = 0; i < 9; ++i) V.R[i] = u[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) V.t[i] = u[9 + i];
      eval_cost<MODE>(e, V, reg, rr[sgn], kk);
    }
    Ja[prm] = (rr[0] - rr[1]) * inv2h[prm];
  }
  J[0] = Ja[0];
  J[1] = Ja[1];
  // PlusJacobian of q (xyzw) [EXT]: rows x, y, z, w; columns delta_xyz.  The pass's rotation columns are for
  // R <- Exp(omega) R, omega = 2 delta: half the manifold's (lm_advance doubles them back).
  const double qx = qc[0], qy = qc[1], qz = qc[2], qw = qc[3];
  J[2] = 0.5 * (Ja[2] * qw + Ja[3] * -qz + Ja[4] * qy + Ja[5] * -qx);
  J[3] = 0.5 * (Ja[2] * qz + Ja[3] * qw + Ja[4] * -qx + Ja[5] * -qy);
  J[4] = 0.5 * (Ja[2] * -qy + Ja[3] * qx + Ja[4] * qw + Ja[5] * -qz);
}

// region markers for tools/isa_mix.py --regions (assembly comments; compiled in only on request)
#ifdef PNEC_ISA_MARKS
#define PNEC_MARK(name) asm volatile("; PNEC_MARK " name)
#else
#define PNEC_MARK(name)
#endif

constexpr int SRC_PLANES = 0;  // the batch's SoA planes in HBM (pnec_hip_problem)
constexpr int SRC_AOS = 1;     // the caller's arrays in the reference layout (streaming handle)
// SRC_DUAL (round 5, an A/B form: -DPNEC_SOLVE_DUAL_AB builds + PNEC_SOLVE_DUAL=1; not instantiated otherwise): the batch's planes, TWO one-wavefront solves per block whose LM steps
// run as ONE instruction stream -- after both wavefronts' passes (a barrier) the first wavefront advances solve 0 in its
// lanes 0..3 and solve 1 in lanes 4..7 (lm_advance is per-lane code on an LDS slab; its only cross-lane traffic is inside
// a quad), then a second barrier publishes both candidates.  Per pair of solves and iteration the step's ~400 issue slots
// are paid once instead of twice; the price is two block barriers per iteration and a wavefront that waits while the
// other steps.  Same arithmetic per solve, hence the same bits -- and 8.8 % SLOWER on the benchmark (36.68 -> 33.45 M solves/s):
// the barriers cost more than the shared step saves.  The measured answer to the round-4 review's "one LM step for two
// solves"; NOTES/round-5.md.
constexpr int SRC_DUAL = 2;
template <int MODE, int CPL, int WPP, int LDSK, bool RESIDENT, int SRC = SRC_PLANES>
__global__ __launch_bounds__(kWave *(SRC == SRC_DUAL ? 2 : WPP), (CPL == 8 && LDSK == 0) ? 1 : 2) void lm_solve_kernel(
    const SolveArgs a) {
  static_assert(SRC != SRC_AOS || RESIDENT, "the AoS source is only built for the on-chip-resident geometries");
  constexpr bool DUAL = SRC == SRC_DUAL;
  static_assert(!DUAL || (WPP == 1 && RESIDENT), "the dual form pairs one-wavefront resident solves");
  constexpr int NW = DUAL ? 2 : WPP;   // wavefronts of the block
  constexpr int NC = num_components(MODE);
  constexpr int RCPL = CPL > 8 ? 8 : CPL;          // correspondences per lane resident on chip
  constexpr int TAILK = CPL - RCPL;                // ... and re-read from memory in every pass (one wavefront only)
  static_assert(TAILK == 0 || (TAILK == 4 && WPP == 1 && RESIDENT && SRC != SRC_AOS), "the tail form is (12, 1, 3) on the batch's planes");
  constexpr int REGK = RESIDENT ? RCPL - LDSK : 1;  // correspondences per lane kept in registers
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  // DUAL: block b holds solves 2 b and 2 b + 1 (of the XCD-contiguous order); an odd batch's last block has one
  const int64_t slot_raw = DUAL ? 2 * xcd_contiguous_index(blockIdx.x, (a.n_solves + 1) / 2) + (threadIdx.x >> 6)
                                : xcd_contiguous_index(blockIdx.x, a.n_solves);
  [[maybe_unused]] const bool exists = !DUAL || slot_raw < a.n_solves;
  const int64_t slot = exists ? slot_raw : a.n_solves - 1;   // (a wavefront without a solve shadows the last one and writes nothing)
  const int64_t pair = a.pair_index ? (int64_t)a.pair_index[slot / a.n_hyp] : slot / a.n_hyp;
  const int64_t s = pair * a.n_hyp + slot % a.n_hyp;
  const double *__restrict__ base = nullptr;
  int n;
  int64_t aos0 = 0;
  if constexpr (SRC == SRC_AOS) {
    aos0 = a.aos_offsets[pair];
    n = (int)(a.aos_offsets[pair + 1] - aos0);
  } else {
    base = a.data + a.block_offset[pair];
    n = a.count[pair];
  }
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const pnec_hip_options &o = a.opt;
  const double reg = a.reg;

  unsigned long long t_begin = 0, t_loaded = 0;
  if (a.trace) t_begin = __builtin_amdgcn_s_memtime();

  // One wavefront per solve: its LM state.  Several (WPP > 1): ONE copy, advanced by the first wavefront while
  // the others wait at the block's barrier -- their SIMDs run other solves' wavefronts meanwhile.  (Until round 2
  // every wavefront kept a copy and advanced it identically to save that barrier: WPP x the ~400 instructions
  // of the step per iteration, 18 % of a two-wavefront solve's issue slots, 28 % of an eight-wavefront one's.)
  __shared__ double slab_all[NW][kSlab];
  __shared__ double unif_all[NW][kUnif];
  __shared__ int ist_all[NW][kINumI];
#ifdef PNEC_ADVANCE_ROWS
  __shared__ int gidx_all[1][16 * kGatherInts];  // lm_advance_rows: each lane's row of the tables of sums
#endif
  [[maybe_unused]] __shared__ double xw[2][WPP > 1 ? WPP : 1][kSumSlots];
  [[maybe_unused]] __shared__ double nunif[RESIDENT ? 1 : kNumPoses][12];   // numeric Jacobian: R | t of the perturbed poses
  [[maybe_unused]] __shared__ double ninv2h[8];
  [[maybe_unused]] __shared__ double ldata[LDSK > 0 ? NW : 1][LDSK > 0 ? LDSK : 1][NC][LDSK > 0 ? kWave : 1];
  double *slab = slab_all[WPP > 1 ? 0 : wave];
  double *unif = unif_all[WPP > 1 ? 0 : wave];
  int *ist = ist_all[WPP > 1 ? 0 : wave];
  [[maybe_unused]] int parity = 0;

  // ---- load this lane's correspondences once (coalesced: consecutive lanes, consecutive doubles)
  double d[REGK][NC];
  if constexpr (RESIDENT && SRC == SRC_AOS)
    load_resident_aos<NC, RCPL, REGK>(a.aos_bvs1 + 3 * aos0, a.aos_bvs2 + 3 * aos0, NC >= 12 ? a.aos_covs + 9 * aos0 : nullptr,
                                     NC >= 18 ? a.aos_covs_host + 9 * aos0 : nullptr, n, wave * RCPL * kWave, lane, d,
                                     &ldata[LDSK > 0 ? wave : 0][0][0][0]);
  else if constexpr (RESIDENT)
    load_resident<NC, RCPL, REGK>(base, n, stride, (DUAL ? 0 : wave) * RCPL * kWave, lane, d, &ldata[LDSK > 0 ? wave : 0][0][0][0]);
  // how many of this wavefront's CPL slots hold any correspondence of the pair (wave-uniform): slot k
  // starts at correspondence first + 128 (k / 2) + (k & 1) (load_resident), lane 0 being the first
  [[maybe_unused]] int nslots = 0;
  if constexpr (RESIDENT) {
#pragma unroll
    for (int k = 0; k < RCPL; ++k) nslots += (n > (DUAL ? 0 : wave) * RCPL * kWave + slot_corr<RCPL, REGK>(k, 0)) ? 1 : 0;
  }
  // The tail (TAILK = 4): correspondences 512 .. 767 as the SECOND wavefront of (8, 2, 3) would hold them in its first
  // four register slots -- tail slot t <-> 512 + 128 (t / 2) + 2 lane + (t & 1) -- accumulated from zero in that order,
  // reduced through the same tree and added to the resident part's sums the way the two wavefronts' sums are added.
  // Same operations in the same order: the result is bit for bit (8, 2, 3)'s.  ntail = tail slots that hold a
  // correspondence (wave-uniform; like (8, 2, 3)'s second wavefront, slots 0, 1 are evaluated together and 2, 3 together).
  [[maybe_unused]] int ntail = 0;
  [[maybe_unused]] const char *tail_base = nullptr;  // first tail correspondence of plane 0, in scalar registers
  [[maybe_unused]] size_t tail_plane_bytes = 0;
  if constexpr (TAILK > 0) {
#pragma unroll
    for (int t = 0; t < TAILK; ++t) ntail += (n > RCPL * kWave + 2 * kWave * (t / 2) + (t & 1)) ? 1 : 0;
    // the pair's block and its plane stride are one value per wavefront: say so, and the tail's loads address as
    // scalar base + 32-bit lane offset + immediate
    const unsigned long long b64 = reinterpret_cast<unsigned long long>(base) + (unsigned long long)RCPL * kWave * sizeof(double);
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
    tail_base = reinterpret_cast<const char *>(((unsigned long long)bhi << 32) | blo);
    tail_plane_bytes = (size_t)(unsigned)__builtin_amdgcn_readfirstlane(stride) * sizeof(double);
    ntail = __builtin_amdgcn_readfirstlane(ntail);
  }
  if (a.trace) {
    // make "payload on chip" mean what it says: wait for the loads before stamping
    __builtin_amdgcn_s_waitcnt(0);
    t_loaded = __builtin_amdgcn_s_memtime();
  }

  // ---- PNECCeres::InitValues(q, t): pnec_ceres.cc:182-186.  The start point is the first
  // "candidate"; the loop's first pass evaluates it (Ceres' iteration zero).
  // Everything that is one value per solve runs in a few lanes only (here lane 0, lm_advance on
  // a quad) against the LDS slab: not faster to issue (measured), but one copy of the state and
  // plain per-lane control flow instead of wave-uniform bookkeeping in scalar registers.
#ifdef PNEC_ADVANCE_ROWS
  if (WPP == 1 || wave == 0) gather_index_init(gidx_all[0], lane);
#endif
  if (lane == 0 && (WPP == 1 || wave == 0)) {
    double th, ph;
    const double *t0 = a.hyp_t ? a.hyp_t + 3 * s : a.init_t + 3 * pair;
    angles_from_vec(t0[0], t0[1], t0[2], th, ph);
    double q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = a.init_q[pair * 4 + k];
      slab[kQc + k] = q[k];
    }
    slab[kThetaC] = th;
    slab[kPhiC] = ph;
    pose_uniforms(th, ph, q, unif);
    ist[kIIter] = 0;
    ist[kIFirst] = 1;
    ist[kIReuseDiag] = 0;
    ist[kINumInvalid] = 0;
    ist[kIStepOk] = 1;
    ist[kILast] = o.max_num_iterations <= 0 ? 1 : 0;
    ist[kIPark] = 0;
    if constexpr (DUAL) ist[kITerm] = exists ? -1 : PNEC_HIP_TERM_MAX_ITERATIONS;   // (>= 0: this slot is done -- or was never there)
  }
  const double inv_max_radius = a.inv_max_radius, inv_min_radius = a.inv_min_radius;  // kernel arguments: scalar
  // the LDS slots arrive by DMA (vmcnt-tracked): they must have landed before the first pass reads them
  if constexpr (RESIDENT && LDSK > 0 && SRC != SRC_AOS) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  if constexpr (WPP > 1 || DUAL) __syncthreads();  // the first wavefront's start state is what all of them read

  int term;
  [[maybe_unused]] bool alive = exists;      // DUAL: this wavefront's solve still iterates (wave-uniform)
  int n_full_passes = 0, n_cost_passes = 0;  // wave-uniform (diagnostics: SolveArgs::work)
  for (;;) {
    if (!DUAL || alive) {
    // ---- one fused pass at the candidate: sum r^2, J'r, J'J ------------------------------
    PNEC_MARK("uniforms");
    {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      PassUniforms U;
#pragma unroll
      for (int i = 0; i < 9; ++i) U.R[i] = to_sgpr(unif[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) U.t[i] = to_sgpr(unif[9 + i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) U.bth[i] = to_sgpr(unif[12 + i]);
      // dt/dphi = (-sin th sin ph, sin th cos ph, 0) = (-t_y, t_x, 0): the same products, a sign flip on the scalar
      // side -- four readfirstlanes fewer per pass
      U.bph[0] = -U.t[1];
      U.bph[1] = U.t[0];
      U.bph[2] = 0.0;
      double c[6];
      bool cost_only = false;
      if constexpr (RESIDENT) cost_only = to_sgpr(ist[kILast]) != 0;  // wave-uniform
      n_cost_passes += cost_only ? 1 : 0;
      n_full_passes += cost_only ? 0 : 1;
      if (cost_only) {
        PNEC_MARK("pass_cost");
        if constexpr (RESIDENT) {
          double a0 = 0.0, z = 0.0;
          pass_cost_resident<MODE, REGK, LDSK>(d, &ldata[LDSK > 0 ? wave : 0][0][0][0], nslots, lane, U, reg,
                                               a0, z);
          c[0] = wave_reduce_acc0_row0(a0);  // row 0: the sum; rows 1..3: zero
          // the finite-Jacobian witness as a wave-uniform 0 / NaN in the place of sum 1
          c[1] = __builtin_amdgcn_ballot_w64(!(z == 0.0)) == 0ull ? 0.0 : __builtin_nan("");
          c[2] = c[3] = c[4] = c[5] = 0.0;
          if constexpr (TAILK > 0) {
            if (ntail > 0) {  // wave-uniform
              double a1 = 0.0, z1 = 0.0;
              const char *tb = tail_base;
              const size_t plane_bytes = tail_plane_bytes;
              const unsigned voff = 16u * (unsigned)lane;
              auto tail_cost = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr unsigned imm = 2u * kWave * 8u * (t / 2) + 8u * (t & 1);
                const bool in = RCPL * kWave + 2 * kWave * (t / 2) + 2 * lane < stride;
                double e[NC];
                load_planes_saddr<NC, (int)imm>(e, tb, plane_bytes, voff, in);
                double r, kk;
                eval_cost<MODE>(e, U, reg, r, kk);
                a1 = __builtin_fma(r, r, a1);
                z1 = __builtin_fma(kk, 0.0, z1);
              };
              tail_cost(std::integral_constant<int, 0>{});
              __builtin_amdgcn_sched_barrier(0);
              tail_cost(std::integral_constant<int, 1>{});
              __builtin_amdgcn_sched_barrier(0);
              if (ntail > 2) {
                tail_cost(std::integral_constant<int, 2>{});
                __builtin_amdgcn_sched_barrier(0);
                tail_cost(std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
              }
              c[0] += wave_reduce_acc0_row0(a1);
              c[1] += __builtin_amdgcn_ballot_w64(!(z1 == 0.0)) == 0ull ? 0.0 : __builtin_nan("");
            }
          }
        }
      } else {
        double acc[kNumAcc];
#pragma unroll
        for (int j = 0; j < kNumAcc; ++j) acc[j] = 0.0;
        PNEC_MARK("pass");
        if constexpr (RESIDENT) {
          pass_resident<MODE, REGK, LDSK>(d, &ldata[LDSK > 0 ? wave : 0][0][0][0], nslots, lane, U, reg, acc);
        } else {
          const bool numeric = a.numeric_jacobian != 0;   // (kernel argument: uniform)
          if (numeric) {
            // the candidate was published behind a block barrier: its perturbed poses, made once, behind another
            if (wave == 0) numeric_poses(slab, lane, nunif, ninv2h);
            __syncthreads();
          }
          for (int idx = threadIdx.x; idx < stride; idx += kWave * WPP) {
            double e[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) e[c] = base[(int64_t)c * stride + idx];
            double r, J[5];
            if (numeric) eval_corr_numeric<MODE>(e, U, nunif, ninv2h, slab + kQc, reg, r, J);
            else eval_corr<MODE>(e, U, reg, r, J);
            accumulate(r, J, acc);
          }
        }
        PNEC_MARK("reduce");
        wave_reduce21_rows(acc, c);
        if constexpr (TAILK > 0) {
          if (ntail > 0) {  // wave-uniform
            // the resident part's sums wait in the table (the registers are needed: payload + accumulators + a
            // correspondence in flight fill the file), where the tail's are added to them below
            double *park_sums = slab + kSums + (to_sgpr(ist[kIPark]) ^ 1) * kSumSlots;
            if ((lane & 15) == 0) {
#pragma unroll
              for (int i = 0; i < 6; ++i) park_sums[(lane >> 4) * 6 + i] = c[i];
            }
            // the tail, from memory (L2: it was read moments ago by the previous pass), into fresh accumulators
#pragma unroll
            for (int j = 0; j < kNumAcc; ++j) acc[j] = 0.0;
            // addresses as (scalar plane base) + (one 32-bit per-lane offset) + (an immediate per slot): nothing to
            // keep in 64-bit vector registers across the loop (24 precomputed pointers did not fit and spilled)
            const char *tb = tail_base;
            const size_t plane_bytes = tail_plane_bytes;
            const unsigned voff = 16u * (unsigned)lane;
            auto tail_slot = [&](auto tc) {
              constexpr int t = decltype(tc)::value;
              constexpr unsigned imm = 2u * kWave * 8u * (t / 2) + 8u * (t & 1);
              const bool in = RCPL * kWave + 2 * kWave * (t / 2) + 2 * lane < stride;
              double e[NC];
              load_planes_saddr<NC, (int)imm>(e, tb, plane_bytes, voff, in);
              double r, J[5];
              eval_corr<MODE>(e, U, reg, r, J);
              accumulate(r, J, acc);
            };
            // (one correspondence in flight at a time: the scheduler must not hoist the later slots' loads -- there
            // are no registers for them)
            tail_slot(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            tail_slot(std::integral_constant<int, 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if (ntail > 2) {
              tail_slot(std::integral_constant<int, 2>{});
              __builtin_amdgcn_sched_barrier(0);
              tail_slot(std::integral_constant<int, 3>{});
              __builtin_amdgcn_sched_barrier(0);
            }
            double c2[6];
            wave_reduce21_rows(acc, c2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // resident + tail, the order in which (8, 2, 3) adds its two wavefronts' sums
#pragma unroll
            for (int i = 0; i < 6; ++i) c[i] = park_sums[(lane >> 4) * 6 + i] + c2[i];
          }
        }
      }
      // the four row leaders store the sums they own (sum_slot layout) into the table the current
      // point does not own
      double *cand_sums = slab + kSums + (to_sgpr(ist[kIPark]) ^ 1) * kSumSlots;
      if constexpr (WPP == 1) {
        const bool fin = rows_all_finite(c);
        if ((lane & 15) == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) cand_sums[(lane >> 4) * 6 + i] = c[i];
          if (lane == 0) slab[kSumsFinite] = fin ? 1.0 : 0.0;
        }
      } else {
        if ((lane & 15) == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) xw[parity][wave][(lane >> 4) * 6 + i] = c[i];
        }
        __syncthreads();
        if (wave == 0 && lane == 0) {
          double z = 0.0;
#pragma unroll
          for (int j = 0; j < kSumSlots; ++j) {
            double t = xw[parity][0][j];
#pragma unroll
            for (int w = 1; w < WPP; ++w) t += xw[parity][w][j];
            cand_sums[j] = t;
            z = __builtin_fma(t, 0.0, z);
          }
          slab[kSumsFinite] = (z == 0.0) ? 1.0 : 0.0;
        }
        parity ^= 1;
      }
    }

    }  // (DUAL: the wavefront of a finished solve only keeps the barriers' count)

    // ---- accept / reject, trust region, next candidate: one lane
    PNEC_MARK("advance");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    int t = -1;
    if constexpr (DUAL) {
      __syncthreads();   // both solves' sums are in their slabs
      if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        if (lane < 8) {   // lanes 0..3: solve 0, lanes 4..7: solve 1 -- one instruction stream for both steps
          const int sv = lane >> 2;
          if (ist_all[sv][kITerm] < 0) {
            t = lm_advance<RESIDENT && kCostFirst>(slab_all[sv], ist_all[sv], unif_all[sv], o, inv_max_radius, inv_min_radius);
            if ((lane & 3) == 0) ist_all[sv][kITerm] = t;
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();   // the next candidates (or the verdicts) are published
      // both wavefronts read both verdicts (written before the barrier): the loop ends for both in the same trip
      const int t0_ = to_sgpr(ist_all[0][kITerm]), t1_ = to_sgpr(ist_all[1][kITerm]);
      term = wave == 0 ? t0_ : t1_;
      if (term >= 0) alive = false;
      if (t0_ >= 0 && t1_ >= 0) break;
      continue;
    }
    // the chain below is latency-bound: let it win the issue arbitration against the pass of the
    // other wavefront on this SIMD, which has independent work to fill the gaps (+1.2 %)
    if constexpr (WPP == 1) {
      __builtin_amdgcn_s_setprio(3);
#ifndef PNEC_ADVANCE_ROWS
      if (lane < 4) t = lm_advance<RESIDENT && kCostFirst>(slab, ist, unif, o, inv_max_radius, inv_min_radius);  // one quad, identical work (see the sincos exchange)
#else
      t = lm_advance_rows(slab, ist, unif, gidx_all[0], o, inv_max_radius, inv_min_radius, lane);  // every lane: one component per lane
#endif
      term = to_sgpr(t);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    } else {
      if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
#ifndef PNEC_ADVANCE_ROWS
        if (lane < 4) t = lm_advance<RESIDENT && kCostFirst>(slab, ist, unif, o, inv_max_radius, inv_min_radius);
#else
        t = lm_advance_rows(slab, ist, unif, gidx_all[0], o, inv_max_radius, inv_min_radius, lane);
#endif
        if (lane == 0) ist[kITerm] = t;
        __builtin_amdgcn_s_setprio(0);
      }
      __syncthreads();  // the next candidate (or the verdict) is published: everybody reads it
      term = to_sgpr(ist[kITerm]);
    }
    if (term >= 0) break;
  }

  PNEC_MARK("result");
  if (DUAL ? (lane == 0 && exists) : threadIdx.x == 0) {
    write_result(a, s, slab, ist[kIIter], term);
    if (a.work) {
      atomicAdd(a.work + 0, (unsigned long long)n_full_passes * (unsigned long long)n);
      atomicAdd(a.work + 1, (unsigned long long)n_cost_passes * (unsigned long long)n);
    }
    if constexpr (SRC == SRC_AOS) {
      // results live in pinned host memory: make them visible system-wide, then count this block in;
      // the block that completes the submit raises the host's flag (the host polls it, no stream sync)
      __threadfence_system();
      const unsigned long long done = atomicAdd(a.done_counter, 1ull) + 1ull;
      if (done == a.n_blocks_total) {
        __hip_atomic_store(a.host_flag, a.flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (a.trace && threadIdx.x == 0) {
      unsigned hw = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      a.trace[4 * (size_t)blockIdx.x + 0] = t_begin;
      a.trace[4 * (size_t)blockIdx.x + 1] = t_loaded;
      a.trace[4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
      a.trace[4 * (size_t)blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hw;
    }
  }
}

}  // namespace pnec_hip
