// pnec_stream.inl -- part of pnec_capi.hip (inside extern "C"): the streaming handle.
//
// The reference's odometry calls PNEC::Solve / PNECCeres::Optimize once per frame
// (src/rel_pose_estimation/frame2frame.cc:122-141, src/pnec_vo.cc:220-261).  A batch object per call
// (hipMalloc, three blocking copies, a pack launch, a solve launch, five blocking copies, hipFree) costs
// more than the CPU needs for the whole solve, so the per-frame path gets a persistent handle instead:
//   * `slots` staging slots in PINNED, device-mapped host memory, allocated once;
//   * submit = memcpy of the caller's arrays into a slot + ONE kernel launch: the solve kernel reads the
//     reference-layout arrays over PCIe itself (SRC_AOS loader), keeps the pair on chip for the whole LM
//     loop as usual, writes the result record back into the slot and raises the slot's flag;
//   * wait = the host polling that flag (no stream synchronisation, no copy-back);
//   * several submits may be in flight (a micro-batching window of `slots`), and one submit may carry
//     several pairs (one workgroup each).
// Pairs too large for the register-resident geometries take the staged route through a capacity-shaped batch
// owned by the handle (re-shaped to the submit's sizes, pack, solve into the slot: nothing allocated per submit
// once the batch exists).
struct pnec_hip_stream {
  int device = 0;
  int32_t max_corr = 0;   // correspondences per submit
  int32_t max_pairs = 0;  // pairs per submit
  int32_t slots = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  struct Slot {
    // pinned + mapped; h_* host addresses, d_* the same memory as the device sees it
    char *h_base = nullptr, *d_base = nullptr;
    size_t bytes = 0;
    int64_t ticket = 0;      // ticket in flight in this slot (0: free)
    int64_t n_pairs = 0;
    unsigned long long blocks_done = 0;  // what the slot's device counter reads once every submit so far is done
    bool staged = false;     // went through the staged route: completion is `ev`
    hipEvent_t ev = nullptr;
  };
  std::vector<Slot> slot;
  unsigned long long *d_counters = nullptr;  // one completion counter per slot
  int64_t next_ticket = 1;
  // layout of a slot (offsets in bytes)
  size_t o_flag, o_offsets, o_pairidx, o_q, o_t, o_b1, o_b2, o_cv, o_ch, o_oq, o_ot, o_oc, o_oi, o_os;
  // staged route
  pnec_hip_problem *big = nullptr;
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int pnec_hip_stream_create(int device, int32_t max_corr, int32_t max_pairs, int32_t slots, void *stream_,
                           pnec_hip_stream **out) {
  if (!out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (max_corr < 1 || max_pairs < 1 || slots < 1 || slots > 64)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need max_corr >= 1, max_pairs >= 1, 1 <= slots <= 64");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed (no such device?)");
  pnec_hip_stream *s = new (std::nothrow) pnec_hip_stream();
  if (!s) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
  s->device = device;
  s->max_corr = max_corr;
  s->max_pairs = max_pairs;
  s->slots = slots;
  auto bail = [&](hipError_t e, const char *what) {
    pnec_hip_stream_destroy(s);
    return fail_hip(e, what);
  };
  if (stream_) {
    s->stream = (hipStream_t)stream_;
    s->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) return bail(e, "hipStreamCreate");
  }
  const size_t M = (size_t)max_corr, P = (size_t)max_pairs;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 64); return at; };
  s->o_flag = take(64);
  s->o_offsets = take(sizeof(int64_t) * (P + 1));
  s->o_pairidx = take(sizeof(int32_t) * P);
  s->o_q = take(sizeof(double) * 4 * P);
  s->o_t = take(sizeof(double) * 3 * P);
  s->o_b1 = take(sizeof(double) * 3 * M);
  s->o_b2 = take(sizeof(double) * 3 * M);
  s->o_cv = take(sizeof(double) * 9 * M);
  s->o_ch = take(sizeof(double) * 9 * M);
  s->o_oq = take(sizeof(double) * 4 * P);
  s->o_ot = take(sizeof(double) * 3 * P);
  s->o_oc = take(sizeof(double) * P);
  s->o_oi = take(sizeof(int32_t) * P);
  s->o_os = take(sizeof(int32_t) * P);
  const size_t bytes = align_up(o, 4096);
  s->slot.resize((size_t)slots);
  for (auto &sl : s->slot) {
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return bail(e, "hipHostMalloc(slot)");
    sl.h_base = (char *)h;
    sl.bytes = bytes;
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) return bail(e, "hipHostGetDevicePointer");
    sl.d_base = (char *)d;
    std::memset(h, 0, 64);
    e = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
    if (e != hipSuccess) return bail(e, "hipEventCreate");
  }
  hipError_t e = dev_alloc(&s->d_counters, sizeof(unsigned long long) * (size_t)slots);
  if (e == hipSuccess) e = hipMemset(s->d_counters, 0, sizeof(unsigned long long) * (size_t)slots);
  if (e != hipSuccess) return bail(e, "completion counters");
  *out = s;
  return 0;
}

int pnec_hip_stream_destroy(pnec_hip_stream *s) {
  if (!s) return 0;
  DeviceGuard guard(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (auto &sl : s->slot) {
    if (sl.h_base) (void)hipHostFree(sl.h_base);
    if (sl.ev) (void)hipEventDestroy(sl.ev);
  }
  if (s->d_counters) (void)dev_free(s->d_counters);
  if (s->big) pnec_hip_problem_destroy(s->big);
  if (s->stream && s->own_stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return 0;
}

// the ladder entry that holds a pair of n correspondences (same choice the batch path makes)
static bool aos_geometry_for(int mode, int n, Geometry *g) {
  const int (*order)[3];
  const int count = geometry_ladder(mode, &order, /*planes*/ false);
  for (int i = 0; i < count; ++i)
    if ((int64_t)kWave * order[i][0] * order[i][1] >= std::max(n, 1)) {
      *g = {order[i][0], order[i][1], order[i][2], true};
      return true;
    }
  return false;
}

// After a failed launch or a failed wait the slot's device counter and the host's idea of it may have drifted
// apart (some workgroups of a multi-launch submit counted in, others never ran).  Bring both back to zero: drain
// the stream so that nothing still counts, clear the counter, forget the ticket.  The slot is usable again; the
// error that brought us here is what the caller sees.
static void stream_slot_reset(pnec_hip_stream *s, pnec_hip_stream::Slot &sl) {
  const size_t i = (size_t)(&sl - s->slot.data());
  (void)hipStreamSynchronize(s->stream);
  (void)hipGetLastError();
  (void)hipMemset(s->d_counters + i, 0, sizeof(unsigned long long));
  sl.blocks_done = 0;
  sl.ticket = 0;
  __atomic_store_n((unsigned long long *)(sl.h_base + s->o_flag), 0ull, __ATOMIC_RELEASE);
}

static int stream_slot_wait(pnec_hip_stream *s, pnec_hip_stream::Slot &sl) {
  if (sl.ticket == 0) return 0;
  if (sl.staged) {
    PNEC_HIP_TRY(hipEventSynchronize(sl.ev));
  } else {
    volatile unsigned long long *flag = (volatile unsigned long long *)(sl.h_base + s->o_flag);
    // the kernel raises the flag with a system-scope release store after its results are out;
    // poll it, yielding to the driver every so often in case the launch itself needs host progress
    unsigned long long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)sl.ticket) {
      if ((++spins & 0xfffff) == 0) {
        const hipError_t q = hipStreamQuery(s->stream);
        if (q != hipSuccess && q != hipErrorNotReady) return fail_hip(q, "streaming solve");
        if (q == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != (unsigned long long)sl.ticket)
          return fail(PNEC_HIP_ERR_HIP_RUNTIME, "streaming solve finished without raising its flag");
      }
    }
  }
  return 0;
}

int pnec_hip_stream_submit(pnec_hip_stream *s, int mode, int64_t n_pairs, const int64_t *offsets,
                           const double *bvs1, const double *bvs2, const double *covs, const double *covs_host,
                           const double *init_q, const double *init_t, double reg, const pnec_hip_options *opt_in,
                           int64_t *ticket) {
  if (!s || !ticket || !offsets || !init_q || !init_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  *ticket = 0;
  if (mode < PNEC_HIP_MODE_NEC || mode > PNEC_HIP_MODE_SYM) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown mode");
  if (n_pairs < 1 || n_pairs > s->max_pairs) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "n_pairs outside [1, max_pairs]");
  if (offsets[0] != 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  int n_big = 0;
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t n = offsets[p + 1] - offsets[p];
    if (n < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
    n_big = (int)std::max<int64_t>(n_big, n);
  }
  const int64_t M = offsets[n_pairs];
  if (M > s->max_corr) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "more correspondences than the handle was created for");
  const int nc = num_components(mode);
  if (M > 0 && (!bvs1 || !bvs2 || (nc >= 12 && !covs) || (nc >= 18 && !covs_host)))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "a bearing / covariance array the residual family needs is NULL");
  pnec_hip_options opt;
  if (opt_in) opt = *opt_in; else pnec_hip_default_options(&opt);
  if (opt.max_num_iterations < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "max_num_iterations < 0");
  if (opt.flags & ~(PNEC_HIP_OPT_COUNT_PASSES | PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pnec_hip_options.flags: undefined bit set");
  if (opt.flags & PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "the numeric-Jacobian verification mode is a pnec_hip_solve option "
                                          "(the streaming handle's AoS-source kernels are the resident forms only)");
  // the handle picks each pair's geometry from the auto-tuner's ladder (the only AoS-source kernels built), so a
  // forced geometry cannot be honoured here: refuse it rather than silently run something else
  if (opt.corr_per_lane != 0 || opt.waves_per_pair != 0 || opt.lds_corr_per_lane != 0)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "forced launch tunings (corr_per_lane / waves_per_pair / lds_corr_per_lane) "
                                          "are not available on the streaming handle");
  DeviceGuard guard(s->device);

  const int64_t t = s->next_ticket;
  pnec_hip_stream::Slot &sl = s->slot[(size_t)(t % s->slots)];
  if (sl.ticket != 0)  // the ring is full of results nobody has collected: dropping the oldest is not ours to decide
    return fail(PNEC_HIP_ERR_BUSY, "every slot holds an uncollected ticket: pnec_hip_stream_wait the oldest first");

  // ---- stage the arguments (the caller's buffers are free again when this returns)
  char *h = sl.h_base;
  std::memcpy(h + s->o_offsets, offsets, sizeof(int64_t) * (size_t)(n_pairs + 1));
  std::memcpy(h + s->o_q, init_q, sizeof(double) * 4 * (size_t)n_pairs);
  std::memcpy(h + s->o_t, init_t, sizeof(double) * 3 * (size_t)n_pairs);
  if (M > 0) {
    std::memcpy(h + s->o_b1, bvs1, sizeof(double) * 3 * (size_t)M);
    std::memcpy(h + s->o_b2, bvs2, sizeof(double) * 3 * (size_t)M);
    if (nc >= 12) std::memcpy(h + s->o_cv, covs, sizeof(double) * 9 * (size_t)M);
    if (nc >= 18) std::memcpy(h + s->o_ch, covs_host, sizeof(double) * 9 * (size_t)M);
  }
  char *d = sl.d_base;
  Geometry gmax;
  const bool resident = aos_geometry_for(mode, n_big, &gmax);
  sl.staged = !resident;
  sl.n_pairs = n_pairs;
  if (resident) {
    SolveArgs a;
    std::memset(&a, 0, sizeof(a));
    a.aos_bvs1 = (const double *)(d + s->o_b1);
    a.aos_bvs2 = (const double *)(d + s->o_b2);
    a.aos_covs = (const double *)(d + s->o_cv);
    a.aos_covs_host = (const double *)(d + s->o_ch);
    a.aos_offsets = (const int64_t *)(d + s->o_offsets);
    a.init_q = (const double *)(d + s->o_q);
    a.init_t = (const double *)(d + s->o_t);
    a.out_q = (double *)(d + s->o_oq);
    a.out_t = (double *)(d + s->o_ot);
    a.out_cost = (double *)(d + s->o_oc);
    a.out_iterations = (int32_t *)(d + s->o_oi);
    a.out_status = (int32_t *)(d + s->o_os);
    a.done_counter = s->d_counters + (t % s->slots);
    a.host_flag = (unsigned long long *)(d + s->o_flag);
    a.flag_value = (unsigned long long)t;
    // the slot's counter only ever counts up (no reset launch between submits): this submit is complete
    // when it reaches the number of workgroups ever launched on the slot
    // (the host's count advances only once every launch of this submit has been accepted)
    const unsigned long long blocks_total = sl.blocks_done + (unsigned long long)n_pairs;
    a.n_blocks_total = blocks_total;
    a.n_hyp = 1;
    a.reg = reg;
    a.opt = opt;
    finish_args(a);
    auto launch = [&](const Geometry &g, const SolveArgs &aa) -> hipError_t {
      switch (mode) {
        case PNEC_HIP_MODE_NEC: return launch_solve_aos_mode_0(g.cpl, g.wpp, g.ldsk, aa, s->stream);
        case PNEC_HIP_MODE_TARGET: return launch_solve_aos_mode_1(g.cpl, g.wpp, g.ldsk, aa, s->stream);
        case PNEC_HIP_MODE_HOST: return launch_solve_aos_mode_2(g.cpl, g.wpp, g.ldsk, aa, s->stream);
        default: return launch_solve_aos_mode_3(g.cpl, g.wpp, g.ldsk, aa, s->stream);
      }
    };
    hipError_t e = hipSuccess;
    if (n_pairs == 1) {
      a.n_solves = 1;
      e = launch(gmax, a);
    } else {
      // one launch per geometry in use, over the pairs that geometry serves -- the same grouping the
      // batch path makes, so every pair runs the kernel it would run there
      int32_t *idx = (int32_t *)(h + s->o_pairidx);
      std::vector<Geometry> gs((size_t)n_pairs);
      for (int64_t p = 0; p < n_pairs; ++p) aos_geometry_for(mode, (int)(offsets[p + 1] - offsets[p]), &gs[(size_t)p]);
      std::vector<char> done((size_t)n_pairs, 0);
      int64_t filled = 0;
      for (int64_t p = 0; p < n_pairs && e == hipSuccess; ++p) {
        if (done[(size_t)p]) continue;
        const Geometry g = gs[(size_t)p];
        const int64_t first = filled;
        for (int64_t q = p; q < n_pairs; ++q)
          if (!done[(size_t)q] && gs[(size_t)q].cpl == g.cpl && gs[(size_t)q].wpp == g.wpp && gs[(size_t)q].ldsk == g.ldsk) {
            idx[filled++] = (int32_t)q;
            done[(size_t)q] = 1;
          }
        SolveArgs ab = a;
        ab.pair_index = (const int32_t *)(d + s->o_pairidx) + first;
        ab.n_solves = filled - first;
        e = launch(g, ab);
      }
    }
    if (e != hipSuccess) {
      // some launches of this submit may be running and will count into the slot: drain and re-zero it
      const int rc = fail_hip(e, "streaming solve launch");
      const std::string msg = g_last_error;
      stream_slot_reset(s, sl);
      g_last_error = msg;
      return rc;
    }
    sl.blocks_done = blocks_total;
  } else {
    // staged route: a persistent batch of the handle at the handle's capacity, re-shaped (no allocation, no
    // device-wide drain) to this submit's sizes; it is re-made only when the residual family changes
    if (s->big && s->big->mode != mode) {
      pnec_hip_problem_destroy(s->big);
      s->big = nullptr;
    }
    if (!s->big)
      if (int rc = pnec_hip_problem_create_capacity(s->device, mode, s->max_pairs, s->max_corr, &s->big)) return rc;
    if (int rc = pnec_hip_problem_reshape(s->big, n_pairs, offsets, s->stream)) return rc;
    if (int rc = pnec_hip_problem_fill(s->big, 0, n_pairs, (const double *)(d + s->o_b1), (const double *)(d + s->o_b2),
                                       nc >= 12 ? (const double *)(d + s->o_cv) : nullptr,
                                       nc >= 18 ? (const double *)(d + s->o_ch) : nullptr, PNEC_HIP_MEM_DEVICE, s->stream))
      return rc;
    if (int rc = pnec_hip_solve(s->big, (const double *)(d + s->o_q), (const double *)(d + s->o_t), 1, nullptr, reg, &opt,
                                (double *)(d + s->o_oq), (double *)(d + s->o_ot), (double *)(d + s->o_oc),
                                (int32_t *)(d + s->o_oi), (int32_t *)(d + s->o_os), PNEC_HIP_MEM_DEVICE, s->stream))
      return rc;
    PNEC_HIP_TRY(hipEventRecord(sl.ev, s->stream));
  }
  sl.ticket = t;
  s->next_ticket = t + 1;
  *ticket = t;
  return 0;
}

static pnec_hip_stream::Slot *stream_find(pnec_hip_stream *s, int64_t ticket) {
  if (!s || ticket <= 0) return nullptr;
  pnec_hip_stream::Slot &sl = s->slot[(size_t)(ticket % s->slots)];
  return sl.ticket == ticket ? &sl : nullptr;
}

int pnec_hip_stream_poll(pnec_hip_stream *s, int64_t ticket, int32_t *done) {
  if (!s || !done) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  pnec_hip_stream::Slot *sl = stream_find(s, ticket);
  if (!sl) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown or already collected ticket");
  if (sl->staged) {
    DeviceGuard guard(s->device);
    const hipError_t q = hipEventQuery(sl->ev);
    if (q != hipSuccess && q != hipErrorNotReady) return fail_hip(q, "streaming solve");
    *done = q == hipSuccess ? 1 : 0;
  } else {
    volatile unsigned long long *flag = (volatile unsigned long long *)(sl->h_base + s->o_flag);
    *done = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == (unsigned long long)ticket ? 1 : 0;
  }
  return 0;
}

int pnec_hip_stream_wait(pnec_hip_stream *s, int64_t ticket, double *out_q, double *out_t, double *out_cost,
                         int32_t *out_iterations, int32_t *out_status) {
  if (!s) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "stream is NULL");
  pnec_hip_stream::Slot *sl = stream_find(s, ticket);
  if (!sl) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown or already collected ticket");
  DeviceGuard guard(s->device);
  if (int rc = stream_slot_wait(s, *sl)) {
    // the ticket is lost, the slot is not: release it (counter and flag re-zeroed) instead of leaving it BUSY
    const std::string msg = g_last_error;
    stream_slot_reset(s, *sl);
    g_last_error = msg;
    return rc;
  }
  const size_t P = (size_t)sl->n_pairs;
  const char *h = sl->h_base;
  if (out_q) std::memcpy(out_q, h + s->o_oq, sizeof(double) * 4 * P);
  if (out_t) std::memcpy(out_t, h + s->o_ot, sizeof(double) * 3 * P);
  if (out_cost) std::memcpy(out_cost, h + s->o_oc, sizeof(double) * P);
  if (out_iterations) std::memcpy(out_iterations, h + s->o_oi, sizeof(int32_t) * P);
  if (out_status) std::memcpy(out_status, h + s->o_os, sizeof(int32_t) * P);
  sl->ticket = 0;
  return 0;
}
