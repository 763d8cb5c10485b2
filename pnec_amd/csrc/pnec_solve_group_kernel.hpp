// pnec_solve_group_kernel.hpp -- the multi-hypothesis form of the on-device Levenberg-Marquardt kernel (round 6).
//
// pnec_hip_solve with n_hyp > 1 runs several solves on ONE pair's correspondences (random t-hat restarts sharing the
// pair's payload: north_star "random t-hat restarts iterate without host round-trips", BASELINE config 4).  The
// one-solve-per-block kernel (pnec_solve_kernel.hpp) treats each (pair, hypothesis) as a pair of its own: every block
// loads the payload again, and with several wavefronts per solve every pass ends in two block barriers around ONE LM
// step that four lanes of the first wavefront execute while everything else on the CU waits (an 8-wavefront solve owns
// its CU: 144 KB of LDS) -- on config 4 that step, ~10 k clocks of dependent FP64, is more than half of an iteration.
//
// Here a block owns a pair's payload ONCE and walks G of its hypotheses side by side:
//   * pass phase: the wavefronts evaluate the G candidate poses one after the other against the registers / LDS they
//     already hold (each pose: 17 uniforms in SGPRs, the same pass_resident / pass_cost_resident as the one-solve kernel);
//     the per-wavefront sums go to LDS and ONE barrier follows every SB = 2 poses (double-buffered), after which the
//     wavefront that OWNS a pose's hypothesis adds the wavefronts' sums in the fixed order 0 .. WPP-1 -- a group pass costs
//     G / 2 + 1 barriers instead of 2 G;
//   * step phase: G = WPP, and every wavefront owns exactly one hypothesis (owner_wave): all G LM steps run AT THE SAME
//     TIME, one per wavefront, each on its own instruction stream -- the ~500-instruction latency chain is waited for once
//     per G solves, no two hypotheses share a divergent wavefront, and the extra work (sums + step) is the same for every
//     wavefront, so they reach the barriers together.  (First built with all G steps in the quads of the first wavefront:
//     1.35 -> 0.87 ms on config 4; accept / reject / retry paths of different hypotheses serialise there.)
// A hypothesis keeps its own LM state (slab, pass kind, termination) exactly as in the one-solve kernel; finished ones
// are skipped by a wave-uniform branch.  Same per-correspondence arithmetic, same reduction tree, same cross-wavefront
// order, same lm_advance: the results are BIT-IDENTICAL to the one-solve-per-block kernel (tests/test_parity_gpu.py).
//
// G = WPP: what fits beside the payload at the occupancy the geometry was built for (8 wavefronts x 18 KB leave 16 KB of
// the CU's 160 KB: eight hypotheses' state 6.8 KB + the exchange buffer 6 KB).
#pragma once

#include "pnec_solve_kernel.hpp"

namespace pnec_hip {

constexpr int kGroupSub = 2;   // poses between two barriers of the pass phase (the exchange buffer holds 2 x this many)

// the several-wavefront geometries of the auto-tuner's ladders (pnec_capi.hip geometry_ladder)
#define PNEC_FOR_EACH_GROUP_GEOMETRY(X) X(4, 2, 0) X(4, 4, 0) X(4, 8, 0) X(8, 2, 3) X(8, 4, 3) X(8, 8, 3)
__host__ __device__ constexpr bool group_geometry_listed(int cpl, int wpp, int ldsk) {
#define PNEC_GROUP_MATCH(CPL, WPP, LDSK) if (cpl == CPL && wpp == WPP && ldsk == LDSK) return true;
  PNEC_FOR_EACH_GROUP_GEOMETRY(PNEC_GROUP_MATCH)
#undef PNEC_GROUP_MATCH
  return false;
}
__host__ __device__ constexpr bool group_geometry_ok(int mode, int cpl, int wpp, int ldsk) {
  if (!group_geometry_listed(cpl, wpp, ldsk) || !geometry_ok(mode, cpl, wpp, ldsk)) return false;
  const int nc = num_components(mode);
  const long lds = (long)wpp * (ldsk * nc * kWave * 8) + (long)wpp * ((kSlab + kUnif) * 8 + kINumI * 4) +
                   2L * kGroupSub * wpp * kSumSlots * 8;
  return lds <= 160 * 1024;
}

// hypothesis h = 2 sb + j of a group (sub-batch sb, pose j of it) belongs to wavefront sb + (G / 2) j: the two poses of a
// sub-batch are summed by two different wavefronts right behind the sub-batch's barrier, each by the one that steps it
template <int G>
__device__ __forceinline__ constexpr int group_hyp_of_wave(int wave) { return 2 * (wave % (G / 2)) + wave / (G / 2); }

template <int MODE, int CPL, int WPP, int LDSK>
__global__ __launch_bounds__(kWave *WPP, (CPL == 8 && LDSK == 0) ? 1 : 2) void lm_solve_group_kernel(const SolveArgs a) {
  constexpr int G = WPP;               // hypotheses per block
  constexpr int NC = num_components(MODE);
  constexpr int REGK = CPL - LDSK;
  constexpr int SB = kGroupSub;
  static_assert(G % SB == 0, "the pass phase exchanges sums every SB poses");
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
  // block -> (pair, group of hypotheses): the groups of one pair are consecutive logical blocks, i.e. share an XCD's L2
  const int groups_per_pair = (a.n_hyp + G - 1) / G;
  const int64_t n_pairs_here = a.n_solves / a.n_hyp;
  const int64_t logical = xcd_contiguous_index(blockIdx.x, n_pairs_here * groups_per_pair);
  const int64_t pslot = logical / groups_per_pair;
  const int h0 = (int)(logical % groups_per_pair) * G;
  const int nh = (a.n_hyp - h0) < G ? (a.n_hyp - h0) : G;     // hypotheses of this block (the pair's last group may be short)
  const int64_t pair = a.pair_index ? (int64_t)a.pair_index[pslot] : pslot;
  const double *__restrict__ base = a.data + a.block_offset[pair];
  const int n = a.count[pair];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const pnec_hip_options &o = a.opt;
  const double reg = a.reg;

  __shared__ double slab_all[G][kSlab];
  __shared__ double unif_all[G][kUnif];
  __shared__ int ist_all[G][kINumI];
  __shared__ double xw[2][SB][WPP][kSumSlots];
  [[maybe_unused]] __shared__ double ldata[LDSK > 0 ? WPP : 1][LDSK > 0 ? LDSK : 1][NC][LDSK > 0 ? kWave : 1];

  // ---- the pair's payload, once
  double d[REGK][NC];
  load_resident<NC, CPL, REGK>(base, n, stride, wave * CPL * kWave, lane, d, &ldata[LDSK > 0 ? wave : 0][0][0][0]);
  int nslots = 0;
#pragma unroll
  for (int k = 0; k < CPL; ++k) nslots += (n > wave * CPL * kWave + slot_corr<CPL, REGK>(k, 0)) ? 1 : 0;

  // ---- PNECCeres::InitValues for every hypothesis (pnec_ceres.cc:182-186): by the wavefront that owns it
  const int hw = group_hyp_of_wave<G>(wave);   // this wavefront's hypothesis
  if (lane == 0) {
    const int h = hw;
    double *slab = slab_all[h];
    int *ist = ist_all[h];
    if (h < nh) {
      const int64_t s = pair * a.n_hyp + h0 + h;
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
      pose_uniforms(th, ph, q, unif_all[h]);
    }
    ist[kIIter] = 0;
    ist[kIFirst] = 1;
    ist[kIReuseDiag] = 0;
    ist[kINumInvalid] = 0;
    ist[kIStepOk] = 1;
    ist[kILast] = o.max_num_iterations <= 0 ? 1 : 0;
    ist[kIPark] = 0;
    ist[kITerm] = h < nh ? -1 : PNEC_HIP_TERM_MAX_ITERATIONS;   // (>= 0: nothing to do for this slot)
  }
  const double inv_max_radius = a.inv_max_radius, inv_min_radius = a.inv_min_radius;
  if constexpr (LDSK > 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA'd LDS slots have landed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __syncthreads();

  int parity = 0;
  int n_full_passes = 0, n_cost_passes = 0;   // summed over the block's hypotheses (wave-uniform)
  for (;;) {
    // ---- pass phase: the candidates of the live hypotheses, one after the other, SB between two barriers
#pragma unroll 1
    for (int sb = 0; sb < G / SB; ++sb) {
#pragma unroll 1
      for (int j = 0; j < SB; ++j) {
        const int h = sb * SB + j;
        if (to_sgpr(ist_all[h][kITerm]) >= 0) continue;      // finished (or never there): wave-uniform
        const double *unif = unif_all[h];
        PassUniforms U;
#pragma unroll
        for (int i = 0; i < 9; ++i) U.R[i] = to_sgpr(unif[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) U.t[i] = to_sgpr(unif[9 + i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) U.bth[i] = to_sgpr(unif[12 + i]);
        U.bph[0] = -U.t[1];
        U.bph[1] = U.t[0];
        U.bph[2] = 0.0;
        double c[6];
        const bool cost_only = to_sgpr(ist_all[h][kILast]) != 0;
        n_cost_passes += cost_only ? 1 : 0;
        n_full_passes += cost_only ? 0 : 1;
        if (cost_only) {
          double a0 = 0.0, z = 0.0;
          pass_cost_resident<MODE, REGK, LDSK>(d, &ldata[LDSK > 0 ? wave : 0][0][0][0], nslots, lane, U, reg, a0, z);
          c[0] = wave_reduce_acc0_row0(a0);
          c[1] = __builtin_amdgcn_ballot_w64(!(z == 0.0)) == 0ull ? 0.0 : __builtin_nan("");
          c[2] = c[3] = c[4] = c[5] = 0.0;
        } else {
          double acc[kNumAcc];
#pragma unroll
          for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
          pass_resident<MODE, REGK, LDSK>(d, &ldata[LDSK > 0 ? wave : 0][0][0][0], nslots, lane, U, reg, acc);
          wave_reduce21_rows(acc, c);
        }
        if ((lane & 15) == 0) {
#pragma unroll
          for (int i = 0; i < 6; ++i) xw[parity][j][wave][(lane >> 4) * 6 + i] = c[i];
        }
      }
      __syncthreads();
      if (wave % (G / 2) == sb && lane < kSumSlots && ist_all[hw][kITerm] < 0) {
        // this wavefront's hypothesis was pose j = wave / (G / 2) of the sub-batch: the wavefronts' sums, added in the
        // fixed order 0 .. WPP-1 (what the one-solve kernel does), into the table its current point does not own
        const int j = wave / (G / 2), slot = lane;
        double t = xw[parity][j][0][slot];
#pragma unroll
        for (int w = 1; w < WPP; ++w) t += xw[parity][j][w][slot];
        slab_all[hw][kSums + (ist_all[hw][kIPark] ^ 1) * kSumSlots + slot] = t;
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(!(__builtin_fma(t, 0.0, 0.0) == 0.0));   // the 24 lanes vote
        if (slot == 0) slab_all[hw][kSumsFinite] = bad == 0ull ? 1.0 : 0.0;
      }
      parity ^= 1;
    }

    // ---- step phase: every wavefront advances its own hypothesis (one quad, identical work: see lm_advance)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < 4 && ist_all[hw][kITerm] < 0) {
      const int t = lm_advance<kCostFirst>(slab_all[hw], ist_all[hw], unif_all[hw], o, inv_max_radius, inv_min_radius);
      if (lane == 0) ist_all[hw][kITerm] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __syncthreads();   // the next candidates (or the verdicts) are published
    bool going = false;
#pragma unroll
    for (int h = 0; h < G; ++h) going = going || to_sgpr(ist_all[h][kITerm]) < 0;
    if (!going) break;
  }

  if (lane == 0 && hw < nh)
    write_result(a, pair * a.n_hyp + h0 + hw, slab_all[hw], ist_all[hw][kIIter], ist_all[hw][kITerm]);
  if (threadIdx.x == 0 && a.work) {
    atomicAdd(a.work + 0, (unsigned long long)n_full_passes * (unsigned long long)n);
    atomicAdd(a.work + 1, (unsigned long long)n_cost_passes * (unsigned long long)n);
  }
}

// ---- one wavefront per pair (N <= 512 ...): TWO hypotheses of the pair per wavefront -------------------------------------
// The same idea where a solve is ONE wavefront (the (1..8, 1, *) geometries): the wavefront keeps the pair's payload and
// walks two of its hypotheses -- two passes against the resident data, then BOTH LM steps at once, hypothesis h in quad h
// (lm_advance is per-lane code whose only cross-lane traffic stays inside a quad) -- so the ~400 issue slots of a step are
// paid once per two solves, with no barrier at all: what round 5's SRC_DUAL form (two solves of DIFFERENT pairs per block)
// bought with two block barriers per iteration and lost to them.  Two is what fits: the second hypothesis' state (880 B)
// still leaves eight wavefronts per CU beside the (8, 1, 3) payload (8 x 20 192 B <= 160 KB; a third would not).
// Bit-identical to one solve per block.
constexpr int kPairHyp = 2;
#define PNEC_FOR_EACH_PAIRHYP_GEOMETRY(X) X(1, 1, 0) X(2, 1, 0) X(4, 1, 0) X(8, 1, 3) X(8, 1, 0)
__host__ __device__ constexpr bool pairhyp_geometry_ok(int mode, int cpl, int wpp, int ldsk) {
  if (wpp != 1 || !geometry_ok(mode, cpl, wpp, ldsk)) return false;
#define PNEC_PH_MATCH(CPL, WPP, LDSK) if (cpl == CPL && ldsk == LDSK) return true;
  PNEC_FOR_EACH_PAIRHYP_GEOMETRY(PNEC_PH_MATCH)
#undef PNEC_PH_MATCH
  return false;
}

template <int MODE, int CPL, int LDSK>
__global__ __launch_bounds__(kWave, (CPL == 8 && LDSK == 0) ? 1 : 2) void lm_solve_pairhyp_kernel(const SolveArgs a) {
  constexpr int G = kPairHyp;
  constexpr int NC = num_components(MODE);
  constexpr int REGK = CPL - LDSK;
  const int lane = threadIdx.x;
  const int groups_per_pair = (a.n_hyp + G - 1) / G;
  const int64_t n_pairs_here = a.n_solves / a.n_hyp;
  const int64_t logical = xcd_contiguous_index(blockIdx.x, n_pairs_here * groups_per_pair);
  const int64_t pslot = logical / groups_per_pair;
  const int h0 = (int)(logical % groups_per_pair) * G;
  const int nh = (a.n_hyp - h0) < G ? (a.n_hyp - h0) : G;
  const int64_t pair = a.pair_index ? (int64_t)a.pair_index[pslot] : pslot;
  const double *__restrict__ base = a.data + a.block_offset[pair];
  const int n = a.count[pair];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const pnec_hip_options &o = a.opt;
  const double reg = a.reg;

  __shared__ double slab_all[G][kSlab];
  __shared__ double unif_all[G][kUnif];
  __shared__ int ist_all[G][kINumI];
  [[maybe_unused]] __shared__ double ldata[LDSK > 0 ? LDSK : 1][NC][LDSK > 0 ? kWave : 1];

  double d[REGK][NC];
  load_resident<NC, CPL, REGK>(base, n, stride, 0, lane, d, &ldata[0][0][0]);
  int nslots = 0;
#pragma unroll
  for (int k = 0; k < CPL; ++k) nslots += (n > slot_corr<CPL, REGK>(k, 0)) ? 1 : 0;

  if ((lane & 3) == 0 && (lane >> 2) < G) {
    const int h = lane >> 2;
    double *slab = slab_all[h];
    int *ist = ist_all[h];
    if (h < nh) {
      const int64_t s = pair * a.n_hyp + h0 + h;
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
      pose_uniforms(th, ph, q, unif_all[h]);
    }
    ist[kIIter] = 0;
    ist[kIFirst] = 1;
    ist[kIReuseDiag] = 0;
    ist[kINumInvalid] = 0;
    ist[kIStepOk] = 1;
    ist[kILast] = o.max_num_iterations <= 0 ? 1 : 0;
    ist[kIPark] = 0;
    ist[kITerm] = h < nh ? -1 : PNEC_HIP_TERM_MAX_ITERATIONS;
  }
  const double inv_max_radius = a.inv_max_radius, inv_min_radius = a.inv_min_radius;
  if constexpr (LDSK > 0) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA'd LDS slots have landed
  auto wave_sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  wave_sync();

  int n_full_passes = 0, n_cost_passes = 0;
  for (;;) {
#pragma unroll 1
    for (int h = 0; h < G; ++h) {
      if (to_sgpr(ist_all[h][kITerm]) >= 0) continue;
      const double *unif = unif_all[h];
      PassUniforms U;
#pragma unroll
      for (int i = 0; i < 9; ++i) U.R[i] = to_sgpr(unif[i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) U.t[i] = to_sgpr(unif[9 + i]);
#pragma unroll
      for (int i = 0; i < 3; ++i) U.bth[i] = to_sgpr(unif[12 + i]);
      U.bph[0] = -U.t[1];
      U.bph[1] = U.t[0];
      U.bph[2] = 0.0;
      double c[6];
      const bool cost_only = to_sgpr(ist_all[h][kILast]) != 0;
      n_cost_passes += cost_only ? 1 : 0;
      n_full_passes += cost_only ? 0 : 1;
      if (cost_only) {
        double a0 = 0.0, z = 0.0;
        pass_cost_resident<MODE, REGK, LDSK>(d, &ldata[0][0][0], nslots, lane, U, reg, a0, z);
        c[0] = wave_reduce_acc0_row0(a0);
        c[1] = __builtin_amdgcn_ballot_w64(!(z == 0.0)) == 0ull ? 0.0 : __builtin_nan("");
        c[2] = c[3] = c[4] = c[5] = 0.0;
      } else {
        double acc[kNumAcc];
#pragma unroll
        for (int k = 0; k < kNumAcc; ++k) acc[k] = 0.0;
        pass_resident<MODE, REGK, LDSK>(d, &ldata[0][0][0], nslots, lane, U, reg, acc);
        wave_reduce21_rows(acc, c);
      }
      double *cand_sums = slab_all[h] + kSums + (to_sgpr(ist_all[h][kIPark]) ^ 1) * kSumSlots;
      const bool fin = rows_all_finite(c);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cand_sums[(lane >> 4) * 6 + i] = c[i];
        if (lane == 0) slab_all[h][kSumsFinite] = fin ? 1.0 : 0.0;
      }
    }
    // ---- both steps at once: hypothesis h in quad h
    wave_sync();
    {
      const int h = lane >> 2;
      __builtin_amdgcn_s_setprio(3);
      if (h < G && ist_all[h][kITerm] < 0) {
        const int t = lm_advance<kCostFirst>(slab_all[h], ist_all[h], unif_all[h], o, inv_max_radius, inv_min_radius);
        if ((lane & 3) == 0) ist_all[h][kITerm] = t;
      }
      __builtin_amdgcn_s_setprio(0);
    }
    wave_sync();
    bool going = false;
#pragma unroll
    for (int h = 0; h < G; ++h) going = going || to_sgpr(ist_all[h][kITerm]) < 0;
    if (!going) break;
  }
  if ((lane & 3) == 0 && (lane >> 2) < nh) {
    const int h = lane >> 2;
    write_result(a, pair * a.n_hyp + h0 + h, slab_all[h], ist_all[h][kIIter], ist_all[h][kITerm]);
  }
  if (lane == 0 && a.work) {
    atomicAdd(a.work + 0, (unsigned long long)n_full_passes * (unsigned long long)n);
    atomicAdd(a.work + 1, (unsigned long long)n_cost_passes * (unsigned long long)n);
  }
}

}  // namespace pnec_hip
