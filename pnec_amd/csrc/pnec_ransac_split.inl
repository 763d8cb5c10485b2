// pnec_ransac_split.inl -- part of pnec_frontend.hip (inside namespace pnec_hip, after the one- and two-pair RANSAC kernels).
//
// RANSAC around the eigensolver (pnec.cc:239-272), SPLIT AT THE EIGENVALUE MINIMISATION (round 4).
//
// In the kernels above a wavefront owns one or two pairs for the whole loop, and its Newton phase -- 70 % of what it
// issues -- runs at the pace of the slowest of the minimisations that share it: 333 quad-evaluations on 16 quads take 39
// trips where 21 would do, and a launch ends with the wavefronts of the few pairs that need many rounds (a quarter of it
// is that tail).  Both losses come from tying a hypothesis' minimisation to the wavefront of its pair.  Here it is not:
//
//   ransac_round_kernel   one wavefront per pair and round: [consume the previous round's hypotheses: model, scoring
//                         one model after the other with the exact early drop, the sequential rule] then [prepare the
//                         next round's: sample, 36 sums, jittered start -> a task record in HBM] or, when the pair is
//                         done, [its inliers, their sums, the mask, the compacted inlier batch (InlierExtraction)];
//   es_queue_kernel       persistent wavefronts; each QUAD pulls task after task from one queue over ALL pairs'
//                         hypotheses (next task prefetched into registers while the current one is minimised): every
//                         quad is busy until the queue is dry, whatever its neighbours' problems are.
//
// A round is one launch of each; later rounds may prepare several groups of sixteen hypotheses per pair (everything up
// to the rule's current bound k, capped) -- the rule consumes them in order and stops where it would have stopped, so
// hypotheses beyond the stop are wasted work, never a different result.  Pairs still going after the last round (heavy
// contamination: hundreds of rounds) finish in ransac_eigensolver_kernel<true>, the one-pair kernel resumed from their
// state.  A hypothesis' arithmetic is that of the kernels above, instruction for instruction (same sample, same sums in
// the same order, es_minimise_queue's trip, same model and scoring code): masks, counts, iteration numbers and poses
// are bit for bit theirs (tests/test_chain_scale_gpu.py::test_two_pairs_per_wavefront_ransac_is_bitwise_the_one_pair_form).

constexpr int kTaskD = 48;  // doubles per task record
enum : int { kTkG = 0, kTkV = 36, kTkScale = 39, kTkE = 40, kTkIts = 43, kTkEv1 = 44 };
constexpr int kTaskSel = PNEC_HIP_MAX_RANSAC_SAMPLE;  // ints per task: the sample
constexpr int kQueueChunk = 16;                       // tasks a wavefront claims per atomic

struct RansacPool {
  double *rec;       // [cap][kTaskD]: 36 sums | start (in) / minimiser (out) | gradient scale | eigenvector | iterations | sum f1
  int32_t *sel;      // [cap][kTaskSel]
  int32_t *n_tasks;  // device counter: tasks claimed by the round kernel (may run past cap: the excess was deferred)
  int32_t *queue;    // device counter: the queue kernel's hand-out position
  int32_t cap;
};
// (RansacState, the per-pair state between launches: pnec_frontend.hip, next to RansacArgs)
struct RansacSplitArgs {
  RansacArgs r;
  RansacState st;
  RansacPool prev, next;  // prev: the tasks this launch consumes; next: the ones it prepares
  const int32_t *list_prev, *n_list_prev;  // the pairs this launch works on (null: every pair of the batch)
  int32_t *list_next, *n_list_next;        // pairs with tasks in `next`
  int32_t *list_left, *n_list_left;        // pairs handed to the resumed one-pair kernel
  int32_t cap_round;                       // most hypotheses a pair prepares per round (a multiple of 16)
  int32_t last;                            // no further round: pairs that go on are handed over
};

struct RoundLds {
  int tsel[kHypPerRound][PNEC_HIP_MAX_RANSAC_SAMPLE];
  double models[kHypPerRound][13];   // R, t, kModelCapped
  double best_model[12];
  double G[36];
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// InlierExtraction of one pair by one wavefront (the body of select_kernel): eight 64-chunks at a time, first where
// every kept correspondence goes (mask bytes and ballots only), then the copies component by component
__device__ __forceinline__ void compact_pair(int nc, const double *sb, int n, const uint8_t *mk, double *db, int m, int lane) {
  const int sstride = (n + kWave - 1) & ~(kWave - 1), dstride = (m + kWave - 1) & ~(kWave - 1);
  constexpr int kChunks = 8;
  int written = 0;
  for (int base = 0; base < sstride; base += kChunks * kWave) {
    bool in[kChunks];
    int pos[kChunks];
#pragma unroll
    for (int j = 0; j < kChunks; ++j) {
      const int idx = base + j * kWave + lane;
      in[j] = idx < n && mk[idx] != 0;
      const unsigned long long b = __ballot(in[j]);
      pos[j] = written + __popcll(b & ((1ull << lane) - 1ull));
      written += __popcll(b);
    }
    for (int c = 0; c < nc; ++c) {
      const double *sc = sb + (int64_t)c * sstride + base + lane;
      double *dc = db + (int64_t)c * dstride;
      double v[kChunks];
#pragma unroll
      for (int j = 0; j < kChunks; ++j) v[j] = in[j] ? sc[j * kWave] : 0.0;
#pragma unroll
      for (int j = 0; j < kChunks; ++j)
        if (in[j]) dc[pos[j]] = v[j];
    }
  }
  for (int idx = m + lane; idx < dstride; idx += kWave)  // zero padding of the last 64-chunk
    for (int c = 0; c < nc; ++c) db[(int64_t)c * dstride + idx] = 0.0;
}

// The end of a pair's RANSAC: inliers of the best model (all correspondences when sampling is impossible), the mask,
// their 36 sums and the first inlier (ComposeM on the inlier list starts at its second entry, C7) for
// es_batch_kernel<kEpiTranslation>, and -- with a target -- InlierExtraction (pnec.cc:210-229) into it.
// G: 36 doubles of LDS.  Same arithmetic as the tails of the one- and two-pair kernels.
__device__ __forceinline__ int ransac_finish_pair(const RansacArgs &a, int64_t pair, int n, int stride, const double *base,
                                                  const double (&bR)[9], const double (&bt)[3], bool can_sample, int it,
                                                  int lane, double *G) {
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  int my_count = 0, my_first = 0x7fffffff;
  const int64_t aos0 = a.offsets[pair];
  for (int idx = lane; idx < n; idx += kWave) {
    const double f1[3] = {base[idx], base[(int64_t)stride + idx], base[(int64_t)2 * stride + idx]};
    const double f2[3] = {base[(int64_t)3 * stride + idx], base[(int64_t)4 * stride + idx],
                          base[(int64_t)5 * stride + idx]};
    bool in = true;
    if (can_sample) in = reprojection_score(f1, f2, bR, bt) < a.threshold;
    if (a.out_mask) a.out_mask[aos0 + idx] = in ? 1 : 0;
    if (in) {
      ++my_count;
      if (idx < my_first) my_first = idx;
      const double p[6] = {f2[0] * f2[0], f2[0] * f2[1], f2[0] * f2[2], f2[1] * f2[1], f2[1] * f2[2], f2[2] * f2[2]};
      const double qq[6] = {f1[0] * f1[0], f1[0] * f1[1], f1[0] * f1[2], f1[1] * f1[1], f1[1] * f1[2], f1[2] * f1[2]};
#pragma unroll
      for (int kl = 0; kl < 6; ++kl)
#pragma unroll
        for (int ac = 0; ac < 6; ++ac) acc[6 * kl + ac] = __builtin_fma(p[kl], qq[ac], acc[6 * kl + ac]);
    }
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) {
    const double sres = wave_allreduce_sum(acc[i]);
    if (lane == 0) G[i] = sres;
  }
  const int total = (int)wave_allreduce_sum((double)my_count);
  if (lane == 0) PNEC_WORK_ADD(kWkRansacInlierCorr, n);
  int first = my_first;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(first, off);
    first = o < first ? o : first;
  }
  wave_lds_sync();
  if (lane < 36) a.scratch.G[36 * pair + lane] = G[lane];
  if (lane == 0) {
    double v[3];
    rot_to_cayley(bR, v);
    a.scratch.v0[3 * pair] = v[0]; a.scratch.v0[3 * pair + 1] = v[1]; a.scratch.v0[3 * pair + 2] = v[2];
    a.scratch.n_scale[pair] = (double)(total > 0 ? total : 1);
    a.scratch.first[pair] = total > 0 ? first : -1;
    if (a.out_count) a.out_count[pair] = total;
    if (a.out_iterations) a.out_iterations[pair] = it;
  }
  if (a.sel_data) {  // InlierExtraction: this wavefront's own mask bytes back (each lane reads what it wrote)
    if (lane == 0) {
      a.sel_count[pair] = total;
      if (a.sel_single_offsets) {
        a.sel_single_offsets[0] = 0;
        a.sel_single_offsets[1] = total;
      }
    }
    compact_pair(a.nc, base, n, a.out_mask + aos0, a.sel_data + a.sel_block[pair], total, lane);
  }
  return total;
}

// One group of (up to) sixteen hypotheses of a pair, quad `hyp` on hypothesis it0 + hyp: its sample (registers; copied to
// tsel, LDS, for the writer), the sample's 36 sums and sum of f1 (the quad's lanes split the sample and add up), its
// jittered start.  ransac2_eigensolver_kernel's preparation.  Valid in every lane of an active quad on return.
__device__ __forceinline__ void ransac_prepare_group(const RansacArgs &a, unsigned long long pid, int it0, int nact, int n,
                                                     int st, const double *bs, const double (&v0)[3],
                                                     int (*tsel)[PNEC_HIP_MAX_RANSAC_SAMPLE], int hyp, int role,
                                                     double (&Gl)[36], double (&ev1)[3], double (&vs)[3]) {
  const int ss = a.sample_size;
  const bool active = hyp < nact;
  const unsigned long long hh = (unsigned long long)(it0 + hyp);
  int smp[PNEC_HIP_MAX_RANSAC_SAMPLE];
  if (active) ransac_sample_regs(a.seed, pid, hh, n, ss, smp);
  if (active && role == 0) PNEC_WORK_ADD(kWkRansacHyps, 1);
  ransac_sample_sums(bs, st, ss, active, smp, role, Gl, ev1);
  if (active && role == 0) {
#pragma unroll
    for (int j = 0; j < PNEC_HIP_MAX_RANSAC_SAMPLE; ++j) tsel[hyp][j] = smp[j];
  }
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < 3; ++c) vs[c] = v0[c] + (rng_uniform(a.seed, pid, hh, 1000 + c) - 0.5) * 2.0 * 0.01;
}

// The model of the quad's hypothesis from its minimiser v and eigenvector e: R, and t = +-e signed by the directional
// evidence sum t.(f1 - R f2) over the sample (the quad's lanes split it).  The code of the kernels above.
__device__ __forceinline__ void ransac_model(const double *bs, int st, int ss, bool active, const int *sel /* the sample */,
                                             const double (&v)[3], const double (&e1)[3], int role, double (&R)[9],
                                             double (&t)[3]) {
  cayley_to_rot(v, R);
  double ev = 0.0;
  for (int j = role; j < (active ? ss : 0); j += 4) {
    const int idx = sel[j];
    const double f2[3] = {bs[(int64_t)3 * st + idx], bs[(int64_t)4 * st + idx], bs[(int64_t)5 * st + idx]};
    const double u[3] = {R[0] * f2[0] + R[1] * f2[1] + R[2] * f2[2], R[3] * f2[0] + R[4] * f2[1] + R[5] * f2[2],
                         R[6] * f2[0] + R[7] * f2[1] + R[8] * f2[2]};
    ev -= t[0] * u[0] + t[1] * u[1] + t[2] * u[2];
  }
  ev = (t[0] * e1[0] + t[1] * e1[1] + t[2] * e1[2]) + quad_sum(ev);
  if (ev < 0.0) { t[0] = -t[0]; t[1] = -t[1]; t[2] = -t[2]; }
}

#ifndef PNEC_ROUND_WAVES_PER_SIMD
#define PNEC_ROUND_WAVES_PER_SIMD 2
#endif
template <bool FIRST>
__global__ __launch_bounds__(kWave, PNEC_ROUND_WAVES_PER_SIMD) void ransac_round_kernel(const RansacSplitArgs a) {
  const int lane = threadIdx.x;
  const int hyp = lane >> 2, role = lane & 3;
  __shared__ RoundLds lds;
  const int ss = a.r.sample_size;
  const int64_t n_work = a.list_prev ? (int64_t)*a.n_list_prev : a.r.n_pairs;
  for (int64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
    const int64_t pair = a.list_prev ? (int64_t)a.list_prev[w] : w;
    const int n = a.r.count[pair];
    const int stride = (n + kWave - 1) & ~(kWave - 1);
    const double *base = a.r.data + a.r.block_offset[pair];
    const unsigned long long pid = a.r.pair_id_base + (unsigned long long)pair;
    double q0[4] = {a.r.init_q[4 * pair], a.r.init_q[4 * pair + 1], a.r.init_q[4 * pair + 2], a.r.init_q[4 * pair + 3]};
    {
      const double qn = 1.0 / sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
      for (int c = 0; c < 4; ++c) q0[c] *= qn;
    }
    double R0[9], v0[3];
    rot_from_quat(q0, R0);
    rot_to_cayley(R0, v0);
    const bool can_sample = n >= ss && ss >= 1;
    int it = 0, best_count = -1;
    double k = 1.0;
    bool stop = !can_sample;
    if constexpr (!FIRST) {
      it = a.st.it[pair];
      best_count = a.st.best[pair];
      k = a.st.k[pair];
      const int needed_prev = a.st.needed[pair];
      const int task0 = a.st.task0[pair];
      if (lane < 12) lds.best_model[lane] = a.st.model[12 * pair + lane];
      wave_lds_sync();
      // ---- consume the hypotheses the queue kernel has minimised: group after group of sixteen, quad j building the
      // model of hypothesis j, then one model after the other scored by the whole wavefront (the early drop: a model that
      // cannot beat the best so far leaves after a tile or two), the sequential rule as wave-uniform scalar code
      ScoreTiles tiles;  // the pair's bearings (issue and wait back to back: nothing may stand between the two)
      score_tiles_load(tiles, base, stride, n, lane);
      for (int g0 = 0; g0 < needed_prev && !stop; g0 += kHypPerRound) {
        const int nact = needed_prev - g0 < kHypPerRound ? needed_prev - g0 : kHypPerRound;
        const bool active = hyp < nact;
        const int64_t task = (int64_t)task0 + g0 + (active ? hyp : 0);
        const double *rec = a.prev.rec + task * kTaskD;
        double v[3] = {0, 0, 0}, R[9], t[3] = {0, 0, 1}, e1[3] = {0, 0, 0};
        if (active) {
#pragma unroll
          for (int c = 0; c < 3; ++c) { v[c] = rec[kTkV + c]; t[c] = rec[kTkE + c]; e1[c] = rec[kTkEv1 + c]; }
        }
        ransac_model(base, stride, ss, active, a.prev.sel + task * kTaskSel, v, e1, role, R, t);
        if (active && role == 0) {
#pragma unroll
          for (int i = 0; i < 9; ++i) lds.models[hyp][i] = R[i];
          lds.models[hyp][9] = t[0]; lds.models[hyp][10] = t[1]; lds.models[hyp][11] = t[2];
          lds.models[hyp][kModelCapped] = rec[kTkIts] >= (double)kHypothesisMaxIterations ? 1.0 : 0.0;
        }
        wave_lds_sync();
        int winner = -1;
        for (int j = 0; j < nact; ++j) {
          if (!((double)it < k)) { stop = true; break; }
          double Rj[9], tj[3];
#pragma unroll
          for (int i = 0; i < 9; ++i) Rj[i] = lds.models[j][i];
          tj[0] = lds.models[j][9]; tj[1] = lds.models[j][10]; tj[2] = lds.models[j][11];
          const int cj = lds.models[j][kModelCapped] != 0.0 ? 0 : model_inliers_until_beaten(tiles, base, stride, n, Rj, tj, a.r.threshold, lane, best_count);
          if (cj > best_count) {
            best_count = cj;
            winner = j;
            const double wr = (double)cj / (double)n;
            double p_no = 1.0 - pow_sample(wr, ss);
            p_no = fmax(2.220446049250313e-16, p_no);
            p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
            k = log(1.0 - 0.99) / log(p_no);
          }
          ++it;
          if (it > a.r.max_iterations) { stop = true; break; }
        }
        if (winner >= 0 && lane < 12) lds.best_model[lane] = lds.models[winner][lane];
        wave_lds_sync();
      }
    }
    const bool go = !stop && (double)it < k;
    int needed = 0, task0n = 0;
    bool handed_over = false;
    if (go) {
      // the first round evaluates 16 hypotheses before any bound is known; a later one everything the rule can still
      // consume, up to the round's cap (in floating point: with no inlier yet k is ~2e16 or inf)
      needed = it == 0 ? kHypPerRound : (int)fmin(ceil(k - (double)it), (double)a.cap_round);
      bool fits = false;
      if (!a.last) {
        int t0 = 0;
        if (lane == 0) t0 = atomicAdd(a.next.n_tasks, needed);
        t0 = __builtin_amdgcn_readfirstlane(t0);
        task0n = t0;
        fits = t0 + needed <= a.next.cap;
        if (!fits) {  // the pool is full: what of the claimed range lies inside it is marked empty (the queue skips it)
          for (int i = t0 + lane; i < t0 + needed && i < a.next.cap; i += kWave) a.next.rec[(int64_t)i * kTaskD + kTkScale] = 0.0;
        }
      }
      if (fits) {
        for (int g0 = 0; g0 < needed; g0 += kHypPerRound) {
          const int nact = needed - g0 < kHypPerRound ? needed - g0 : kHypPerRound;
          double Gl[36], ev1[3], vs[3];
          ransac_prepare_group(a.r, pid, it + g0, nact, n, stride, base, v0, lds.tsel, hyp, role, Gl, ev1, vs);
          if (hyp < nact) {
            const int64_t task = (int64_t)task0n + g0 + hyp;
            double *rec = a.next.rec + task * kTaskD;
            if (role == 0) {
#pragma unroll
              for (int i = 0; i < 36; ++i) rec[kTkG + i] = Gl[i];
#pragma unroll
              for (int c = 0; c < 3; ++c) { rec[kTkV + c] = vs[c]; rec[kTkEv1 + c] = ev1[c]; }
              rec[kTkScale] = (double)ss;
            }
            for (int j = role; j < ss; j += 4) a.next.sel[task * kTaskSel + j] = lds.tsel[hyp][j];
          }
          wave_lds_sync();  // (the next group overwrites tsel)
        }
        if (lane == 0) a.list_next[atomicAdd(a.n_list_next, 1)] = (int32_t)pair;
      } else {
        handed_over = true;
        needed = 0;
        if (lane == 0) a.list_left[atomicAdd(a.n_list_left, 1)] = (int32_t)pair;
      }
    }
    if (go) {
      if (lane == 0) {
        a.st.it[pair] = it;
        a.st.best[pair] = best_count;
        a.st.k[pair] = k;
        a.st.needed[pair] = needed;
        a.st.task0[pair] = task0n;
      }
      if constexpr (!FIRST) {
        if (lane < 12) a.st.model[12 * pair + lane] = lds.best_model[lane];
      }
      (void)handed_over;
    } else {
      double bR[9], bt[3] = {0.0, 0.0, 1.0};
#pragma unroll
      for (int i = 0; i < 9; ++i) bR[i] = R0[i];
      if (can_sample) {
#pragma unroll
        for (int i = 0; i < 9; ++i) bR[i] = lds.best_model[i];
        bt[0] = lds.best_model[9]; bt[1] = lds.best_model[10]; bt[2] = lds.best_model[11];
      }
      ransac_finish_pair(a.r, pair, n, stride, base, bR, bt, can_sample, it, lane, lds.G);
    }
    wave_lds_sync();  // (the next pair of a grid-stride launch reuses the LDS)
  }
}

// ---- the queue of eigenvalue minimisations -------------------------------------------------------------------------------
// es_minimise_queue's trip (ONE evaluation for every busy quad, whatever its state) over tasks in HBM: a quad that has
// finished stores its result, arms on the task it PREFETCHED into registers while it was minimising (10 doubles per lane:
// the record's 36 sums, start and gradient scale over the quad's four lanes -> the quad's private LDS slot), and
// prefetches the one after.  Tasks are handed out in chunks of sixteen per atomic, in quad order.  A task with gradient
// scale 0 is an unused slot of the pool and finishes at once.
struct EsQueueArgs {
  RansacPool pool;
};
__global__ __launch_bounds__(kWave, PNEC_RANSAC_WAVES_PER_SIMD) void es_queue_kernel(const EsQueueArgs a) {
  enum : int { kInit = 0, kTrial, kShort, kReeval, kDone };
  const int lane = (int)threadIdx.x, quad = lane >> 2, role = lane & 3;
  __shared__ double Gq[kHypPerRound][40];  // per quad: 36 sums | start | gradient scale
  int T = *a.pool.n_tasks;
  T = T < a.pool.cap ? T : a.pool.cap;
  T = __builtin_amdgcn_readfirstlane(T);
  const double h = 1e-6, inv_h = 1.0 / h;
  double v[3] = {0.0, 0.0, 0.0}, eb[3] = {0.0, 0.0, 1.0};
  double f = 0.0, g[3] = {0.0, 0.0, 0.0}, H[9], d[3] = {0.0, 0.0, 0.0};
  double slope = 0.0, alpha = 1.0, trace_cur = 0.0, n_scale = 1.0;
  bool damped = false;
  int state = kDone, it = 0, ls = 0;
  int cur_task = -1;  // the task this quad is minimising
  bool last_eval = false;
#pragma unroll
  for (int i = 0; i < 9; ++i) H[i] = 0.0;
  double pf[10];
  int pf_task = -1;   // the task whose record sits in pf
#pragma unroll
  for (int i = 0; i < 10; ++i) pf[i] = 0.0;
  int win_next = 0, win_end = 0;  // wave-uniform: the claimed tasks not yet handed to a quad
  bool dry = T <= 0;              // wave-uniform: the queue has nothing left to claim
  // hand a task to each quad whose flag `want` is set (in quad order); -1 when the queue is dry
  auto hand_out = [&](bool want) -> int {
    const unsigned long long wb = __builtin_amdgcn_ballot_w64(want && role == 0);
    int need = __builtin_popcountll(wb), given = 0, mine = -1;
    const int rank = __builtin_popcountll(wb & ((1ull << (lane & ~3)) - 1ull));
    while (need > 0) {  // wave-uniform
      if (win_next == win_end) {
        if (dry) break;
        int b = 0;
        if (lane == 0) b = atomicAdd(a.pool.queue, kQueueChunk);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= T) { dry = true; break; }
        win_next = b;
        win_end = b + kQueueChunk < T ? b + kQueueChunk : T;
      }
      const int take = need < win_end - win_next ? need : win_end - win_next;
      if (want && rank >= given && rank < given + take) mine = win_next + (rank - given);
      win_next += take;
      given += take;
      need -= take;
    }
    return mine;
  };
  auto prefetch = [&](int task) {
    pf_task = task;
    if (task >= 0) {
      const double *r = a.pool.rec + (int64_t)task * kTaskD + 10 * role;
#pragma unroll
      for (int i = 0; i < 10; ++i) pf[i] = r[i];
    }
  };
  // the quad starts on the task in its prefetch registers (cur_task = -1: none, the quad idles)
  auto arm = [&]() {
    cur_task = pf_task;
    if (pf_task >= 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) Gq[quad][10 * role + i] = pf[i];
    }
  };
  auto arm_read = [&]() {  // (after the wavefront's LDS fence)
    if (cur_task >= 0) {
      v[0] = Gq[quad][36]; v[1] = Gq[quad][37]; v[2] = Gq[quad][38];
      n_scale = Gq[quad][39];
      eb[0] = 0.0; eb[1] = 0.0; eb[2] = 1.0;
      f = 0.0; g[0] = g[1] = g[2] = 0.0; d[0] = d[1] = d[2] = 0.0;
      slope = 0.0; alpha = 1.0; trace_cur = 0.0; damped = false;
      it = 0; ls = 0; last_eval = false;
      state = n_scale > 0.0 ? kInit : kDone;  // (an unused slot of the pool: nothing to minimise)
    }
  };
  // every quad's first task, and the one after it
  prefetch(hand_out(true));
  arm();
  wave_lds_sync();
  arm_read();
  prefetch(hand_out(cur_task >= 0));
  [[maybe_unused]] int my_evals = 0;
  for (;;) {
#ifdef PNEC_WORK_COUNT
    if (state != kDone) ++my_evals;
#endif
    if (state != kDone) {
      const double *G = Gq[quad];
      // ---- the point this lane evaluates in this trip (es_minimise_queue's trip, line for line)
      double p[3] = {v[0], v[1], v[2]};
      double a_mine = 0.0;
      if (state == kShort) {
        a_mine = alpha * (role == 0 ? 1.0 : (role == 1 ? 0.5 : (role == 2 ? 0.25 : 0.125)));
#pragma unroll
        for (int c = 0; c < 3; ++c) p[c] = v[c] + a_mine * d[c];
      } else {
        if (state == kTrial) {
#pragma unroll
          for (int c = 0; c < 3; ++c) p[c] = v[c] + d[c];
        }
        p[0] += (role == 1 ? h : 0.0);
        p[1] += (role == 2 ? h : 0.0);
        p[2] += (role == 3 ? h : 0.0);
      }
      double gp[3], Mp[9], ep[3] = {eb[0], eb[1], eb[2]};
      const double fp = es_value_grad<1>(G, p, gp, Mp, ep, state != kInit);
      const double trace_p = Mp[0] + Mp[4] + Mp[8];
      bool at_new_point = false;
      if (state == kShort) {
        const int pass = (ls + role < 40 && fp <= f + 1e-4 * a_mine * slope + 4e-16 * trace_p) ? 1 : 0;
        const int p0 = quad_broadcast<0>(pass), p1 = quad_broadcast<1>(pass), p2 = quad_broadcast<2>(pass),
                  p3 = quad_broadcast<3>(pass);
        if (p0 | p1 | p2 | p3) {
          alpha *= p0 ? 1.0 : (p1 ? 0.5 : (p2 ? 0.25 : 0.125));
          const double smax = alpha * fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
          for (int c = 0; c < 3; ++c) v[c] = v[c] + alpha * d[c];
          ++it;
          last_eval = smax < 1e-12 || it >= kHypothesisMaxIterations;
          state = kReeval;  // the eigenvector AT the new point is wanted: one more evaluation even at the end
        } else {
          alpha *= 0.0625;
          ls += 4;
          if (ls >= 40) state = kDone;
        }
      } else {
        const double fx = quad_broadcast<0>(fp), trace_x = quad_broadcast<0>(trace_p);
        double gx[3], Hx[9], ex[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          gx[r] = quad_broadcast<0>(gp[r]);
          Hx[3 * r + 0] = (quad_broadcast<1>(gp[r]) - gx[r]) * inv_h;
          Hx[3 * r + 1] = (quad_broadcast<2>(gp[r]) - gx[r]) * inv_h;
          Hx[3 * r + 2] = (quad_broadcast<3>(gp[r]) - gx[r]) * inv_h;
          ex[r] = quad_broadcast<0>(ep[r]);
        }
        Hx[1] = Hx[3] = 0.5 * (Hx[1] + Hx[3]);
        Hx[2] = Hx[6] = 0.5 * (Hx[2] + Hx[6]);
        Hx[5] = Hx[7] = 0.5 * (Hx[5] + Hx[7]);
        bool take = true;
        if (state == kTrial) {
          take = fx <= f + 1e-4 * slope + 4e-16 * trace_x;
          if (!take) {  // the gradient judges the full step when the value cannot (see es_minimise_quad)
            const double gmax_old = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
            const double gmax_new = fmax(fabs(gx[0]), fmax(fabs(gx[1]), fabs(gx[2])));
            take = (fx - f) <= 1e-13 * trace_x && gmax_new < gmax_old;
          }
          if (take) {
            const double smax = fmax(fabs(d[0]), fmax(fabs(d[1]), fabs(d[2])));
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = v[c] + d[c];
            ++it;
            if (smax < (damped ? 1e-12 : kHypothesisStepDone) || it >= kHypothesisMaxIterations) state = kDone;
          } else {
            state = kShort;
            alpha = 0.5;
            ls = 1;
          }
        }
        if (take) {
          f = fx;
          trace_cur = trace_x;
#pragma unroll
          for (int i = 0; i < 3; ++i) { g[i] = gx[i]; eb[i] = ex[i]; }
#pragma unroll
          for (int i = 0; i < 9; ++i) H[i] = Hx[i];
          if (last_eval) state = kDone;
          at_new_point = state != kDone;
        }
      }
      if (at_new_point) {
        const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        if (gmax <= fmax(1e-14 * (1.0 + fabs(f)) * n_scale, 1.1e-13 * trace_cur)) {
          state = kDone;
        } else {
          const bool ok = levenberg_direction(H, g, role, d, damped);
          if (ok) {
            slope = d[0] * g[0] + d[1] * g[1] + d[2] * g[2];
            state = kTrial;
          } else {
            state = kDone;
          }
        }
      }
    }
    // ---- quads that have finished: the result goes out, the prefetched task comes in, the next one is prefetched
    const bool fin = state == kDone && cur_task >= 0;
    if (fin) {
      if (role == 0) {
        double *r = a.pool.rec + (int64_t)cur_task * kTaskD;
        r[kTkV] = v[0]; r[kTkV + 1] = v[1]; r[kTkV + 2] = v[2];
        r[kTkE] = eb[0]; r[kTkE + 1] = eb[1]; r[kTkE + 2] = eb[2];
        r[kTkIts] = (double)it;
      }
      arm();
    }
    if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {  // wave-uniform
      wave_lds_sync();
      if (fin) arm_read();
      const int nt = hand_out(fin && cur_task >= 0);
      if (fin) prefetch(cur_task >= 0 ? nt : -1);
    }
    if (__builtin_amdgcn_ballot_w64(state != kDone || cur_task >= 0) == 0ull) break;
  }
#ifdef PNEC_WORK_COUNT
  if (role == 0) PNEC_WORK_ADD(kWkRansacEvals, my_evals);
#endif
}
