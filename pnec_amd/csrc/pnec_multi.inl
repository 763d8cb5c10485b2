// pnec_multi.inl -- part of pnec_capi.hip (inside extern "C"): PNEC::Solve for a batch spread over several GPUs of one
// node from ONE process.
//
// The reference's own fan-out is process-level (scripts/run_simulation.sh:52-67, scripts/parallel_kitti.sh:60-69: one
// process per experiment / sequence); frame pairs are independent, so a batch shards with no data-path exchange.  The
// multi-PROCESS form of that (one rank per GPU, one RCCL gather of the result records) lives in pnec_amd/distributed.py +
// bench.py; this is the form a C or C++ caller of the facade can use without torch or MPI: contiguous ranges of pairs
// balanced by correspondence count (the same rule, pnec_hip_partition), one host thread + one batch + one stream per
// device, every thread writing its range of the caller's result arrays -- one process, so no collective is needed.
// RANSAC draws are a function of (seed, GLOBAL pair index) (pnec_hip_pipeline_options.first_pair_id), so the results do
// not depend on the device list: {0}, {0, 0} and {0, 1, ..., 7} give the same bits.

int pnec_hip_partition(int64_t n_pairs, const int64_t *offsets, int32_t n_parts, int64_t *bounds) {
  if (n_pairs < 0 || n_parts < 1 || !bounds || (n_pairs > 0 && !offsets))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pnec_hip_partition: bad arguments");
  // pnec_amd/distributed.py::partition with weights = correspondences per pair: part r starts at the first pair whose
  // cumulative count (exclusive) reaches total * r / n_parts
  for (int32_t r = 0; r <= n_parts; ++r) bounds[r] = 0;
  if (n_pairs == 0) return 0;
  const double total = (double)(offsets[n_pairs] - offsets[0]);
  int64_t i = 0;
  for (int32_t r = 1; r < n_parts; ++r) {
    const double target = total * (double)r / (double)n_parts;
    while (i < n_pairs && (double)(offsets[i] - offsets[0]) < target) ++i;
    bounds[r] = i;
  }
  bounds[n_parts] = n_pairs;
  for (int32_t r = 1; r <= n_parts; ++r)
    if (bounds[r] < bounds[r - 1]) bounds[r] = bounds[r - 1];
  return 0;
}

// ---- the persistent form: one batch + one stream per listed device, kept alive across calls -------------------------
// create sizes every device's batch for its share of (max_pairs, max_corr) -- contiguous ranges balanced by
// correspondence count never exceed total / n + the largest pair, so each holds max_corr / n + max_pair_corr correspondences
// and max_pairs pairs; fill partitions the caller's batch (pnec_hip_partition), re-shapes the device batches in place and
// uploads every shard from its own host thread; solve / solve_pipeline run the shards side by side and write each pair's
// result into the caller's arrays.  Nothing is allocated after the first call of each kind (the batches' staging and
// scratch buffers persist with them) -- tests count hipMalloc calls through pnec_hip_alloc_counters.
struct pnec_hip_multi {
  int mode = 0;
  int64_t max_pairs = 0, max_corr = 0;
  std::vector<int32_t> devices;
  std::vector<pnec_hip_problem *> probs;
  std::vector<hipStream_t> streams;
  std::vector<int64_t> bounds;   // the current partition: shard r holds pairs [bounds[r], bounds[r + 1])
  std::vector<int64_t> corr0;    // first correspondence of every shard
  int64_t n_pairs = 0;
};

// run fn(shard) for every non-empty shard on a host thread of its own (the calling thread takes the first); exceptions
// (thread creation, allocation) do not escape the C boundary: what was started is joined and an error is returned
static int multi_run(pnec_hip_multi *m, const std::function<int(size_t)> &fn, const char *what) {
  const size_t n = m->devices.size();
  std::vector<int> rcs(n, 0);
  std::vector<std::string> msgs(n);
  auto guarded = [&](size_t d) {
    try {
      rcs[d] = fn(d);
      if (rcs[d]) msgs[d] = g_last_error;
    } catch (const std::exception &e) {
      rcs[d] = PNEC_HIP_ERR_HIP_RUNTIME;
      msgs[d] = e.what();
    } catch (...) {
      rcs[d] = PNEC_HIP_ERR_HIP_RUNTIME;
      msgs[d] = "unknown exception";
    }
  };
  std::vector<std::thread> threads;
  int spawn_failed = 0;
  for (size_t d = 1; d < n; ++d) {
    if (m->bounds[d + 1] <= m->bounds[d]) continue;
    try {
      threads.emplace_back(guarded, d);
    } catch (...) {   // no thread to be had: the calling thread does the shard itself
      spawn_failed = 1;
      guarded(d);
    }
  }
  if (n > 0 && m->bounds[1] > m->bounds[0]) guarded(0);
  for (auto &t : threads) t.join();
  (void)spawn_failed;
  for (size_t d = 0; d < n; ++d)
    if (rcs[d])
      return fail(rcs[d], std::string(what) + ": device " + std::to_string(m->devices[d]) + " (shard " + std::to_string(d) + "): " + msgs[d]);
  return 0;
}

int pnec_hip_multi_destroy(pnec_hip_multi *m) {
  if (!m) return 0;
  for (size_t d = 0; d < m->probs.size(); ++d) {
    if (m->streams[d]) {
      DeviceGuard guard(m->devices[d]);
      (void)hipStreamSynchronize(m->streams[d]);
    }
    if (m->probs[d]) pnec_hip_problem_destroy(m->probs[d]);
    if (m->streams[d]) {
      DeviceGuard guard(m->devices[d]);
      (void)hipStreamDestroy(m->streams[d]);
    }
  }
  delete m;
  return 0;
}

int pnec_hip_multi_create(int32_t n_devices, const int32_t *devices, int mode, int64_t max_pairs, int64_t max_corr,
                          int64_t max_pair_corr, pnec_hip_multi **out) {
  if (!out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (n_devices < 1 || !devices) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need at least one device");
  if (mode < PNEC_HIP_MODE_NEC || mode > PNEC_HIP_MODE_SYM) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown mode");
  if (max_pairs < 1 || max_corr < 0 || max_pair_corr < 0 || max_pair_corr > max_corr)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need max_pairs >= 1, 0 <= max_pair_corr <= max_corr");
  int count = 0;
  PNEC_HIP_TRY(hipGetDeviceCount(&count));
  for (int32_t d = 0; d < n_devices; ++d)
    if (devices[d] < 0 || devices[d] >= count) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "no such device in the list");
  pnec_hip_multi *m = nullptr;
  try {
    m = new pnec_hip_multi();
    m->mode = mode;
    m->max_pairs = max_pairs;
    m->max_corr = max_corr;
    m->devices.assign(devices, devices + n_devices);
    m->probs.assign((size_t)n_devices, nullptr);
    m->streams.assign((size_t)n_devices, nullptr);
    m->bounds.assign((size_t)n_devices + 1, 0);
    m->corr0.assign((size_t)n_devices + 1, 0);
  } catch (...) {
    delete m;
    return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
  }
  // a shard of a partition balanced by correspondence count holds at most total / n + one pair's correspondences
  const int64_t shard_corr = std::min<int64_t>(max_corr, (max_corr + n_devices - 1) / n_devices + max_pair_corr);
  for (int32_t d = 0; d < n_devices; ++d) {
    int rc = pnec_hip_problem_create_capacity(devices[d], mode, max_pairs, shard_corr, &m->probs[(size_t)d]);
    if (!rc) {
      DeviceGuard guard(devices[d]);
      // a stream of the shard's own: shards on the SAME device (a device listed twice) overlap instead of queueing
      // behind each other on the null stream; a stream that cannot be made is an error, not a silent fallback
      const hipError_t e = hipStreamCreateWithFlags(&m->streams[(size_t)d], hipStreamNonBlocking);
      if (e != hipSuccess) rc = fail_hip(e, "hipStreamCreateWithFlags");
    }
    if (rc) {
      const std::string msg = g_last_error;
      pnec_hip_multi_destroy(m);
      return fail(rc, msg);
    }
  }
  *out = m;
  return 0;
}

int32_t pnec_hip_multi_num_devices(const pnec_hip_multi *m) { return m ? (int32_t)m->devices.size() : 0; }
int pnec_hip_multi_bounds(const pnec_hip_multi *m, int64_t *bounds) {
  if (!m || !bounds) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  for (size_t r = 0; r < m->bounds.size(); ++r) bounds[r] = m->bounds[r];
  return 0;
}

int pnec_hip_multi_fill(pnec_hip_multi *m, int64_t n_pairs, const int64_t *offsets, const double *bvs1, const double *bvs2,
                        const double *covs, const double *covs_host) {
  if (!m) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "handle is NULL");
  if (n_pairs < 0 || n_pairs > m->max_pairs || !offsets) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "n_pairs beyond the handle's capacity, or offsets NULL");
  if (offsets[0] != 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  if (offsets[n_pairs] > m->max_corr) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "more correspondences than the handle's capacity");
  if (offsets[n_pairs] > 0 && (!bvs1 || !bvs2)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bvs1/bvs2 is NULL");
  if (offsets[n_pairs] > 0 && m->mode != PNEC_HIP_MODE_NEC && !covs) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs is NULL");
  if (offsets[n_pairs] > 0 && m->mode == PNEC_HIP_MODE_SYM && !covs_host) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs_host is NULL");
  const int32_t n = (int32_t)m->devices.size();
  if (int rc = pnec_hip_partition(n_pairs, offsets, n, m->bounds.data())) return rc;
  m->n_pairs = n_pairs;
  for (int32_t r = 0; r <= n; ++r) m->corr0[(size_t)r] = offsets[m->bounds[(size_t)r]];
  const int rc = multi_run(m, [&](size_t d) -> int {
    const int64_t a = m->bounds[d], cnt = m->bounds[d + 1] - a, c0 = m->corr0[d];
    std::vector<int64_t> off((size_t)cnt + 1);
    for (int64_t p = 0; p <= cnt; ++p) off[(size_t)p] = offsets[a + p] - c0;
    if (int rc = pnec_hip_problem_reshape(m->probs[d], cnt, off.data(), m->streams[d])) return rc;
    if (off[(size_t)cnt] == 0) return 0;
    return pnec_hip_problem_fill(m->probs[d], 0, cnt, bvs1 + 3 * c0, bvs2 + 3 * c0, covs ? covs + 9 * c0 : nullptr,
                                 covs_host ? covs_host + 9 * c0 : nullptr, PNEC_HIP_MEM_HOST, m->streams[d]);
  }, "pnec_hip_multi_fill");
  if (rc) {
    // a shard that could not be shaped or filled (e.g. a pair beyond max_pair_corr outgrew its device's batch) leaves
    // the others with the NEW partition: the handle would index the caller's arrays with bounds its shards do not have.
    // Empty it instead -- every shard zero pairs, bounds zero -- so that a solve after a failed fill is a no-op.
    const std::string msg = g_last_error;
    const int64_t none[1] = {0};
    for (size_t d = 0; d < m->probs.size(); ++d)
      if (m->probs[d]) (void)pnec_hip_problem_reshape(m->probs[d], 0, none, m->streams[d]);
    std::fill(m->bounds.begin(), m->bounds.end(), 0);
    std::fill(m->corr0.begin(), m->corr0.end(), 0);
    m->n_pairs = 0;
    return fail(rc, msg);
  }
  return 0;
}

int pnec_hip_multi_solve(pnec_hip_multi *m, const double *init_q, const double *init_t, int32_t n_hyp, const double *hyp_t,
                         double reg, const pnec_hip_options *opt, double *out_q, double *out_t, double *out_cost,
                         int32_t *out_iterations, int32_t *out_status) {
  if (!m) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "handle is NULL");
  if (!init_q || (!init_t && !hyp_t)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "init_q / init_t is NULL");
  const int64_t H = hyp_t ? std::max<int32_t>(n_hyp, 1) : 1;
  return multi_run(m, [&](size_t d) -> int {
    const int64_t a = m->bounds[d];
    return pnec_hip_solve(m->probs[d], init_q + 4 * a, init_t ? init_t + 3 * a : nullptr, (int32_t)H,
                          hyp_t ? hyp_t + 3 * H * a : nullptr, reg, opt, out_q ? out_q + 4 * H * a : nullptr,
                          out_t ? out_t + 3 * H * a : nullptr, out_cost ? out_cost + H * a : nullptr,
                          out_iterations ? out_iterations + H * a : nullptr, out_status ? out_status + H * a : nullptr,
                          PNEC_HIP_MEM_HOST, m->streams[d]);
  }, "pnec_hip_multi_solve");
}

int pnec_hip_multi_solve_pipeline(pnec_hip_multi *m, const double *init_q, const double *init_t,
                                  const pnec_hip_pipeline_options *opt_in, double *out_q, double *out_t,
                                  uint8_t *out_inlier_mask, int32_t *out_inlier_count) {
  if (!m) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "handle is NULL");
  if (!init_q || !init_t || !out_q || !out_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  pnec_hip_pipeline_options o;
  if (opt_in) o = *opt_in; else pnec_hip_default_pipeline_options(&o);
  return multi_run(m, [&](size_t d) -> int {
    const int64_t a = m->bounds[d], c0 = m->corr0[d];
    pnec_hip_pipeline_options po = o;
    po.first_pair_id = o.first_pair_id + a;   // the shard draws as the pairs it holds, not as pairs 0 .. m-1
    return pnec_hip_solve_pipeline(m->probs[d], init_q + 4 * a, init_t + 3 * a, &po, out_q + 4 * a, out_t + 3 * a,
                                   out_inlier_mask ? out_inlier_mask + c0 : nullptr,
                                   out_inlier_count ? out_inlier_count + a : nullptr, PNEC_HIP_MEM_HOST, m->streams[d]);
  }, "pnec_hip_multi_solve_pipeline");
}

// the one-shot convenience on top of the handle: create, fill, solve_pipeline, destroy
int pnec_hip_solve_pipeline_multi(int32_t n_devices, const int32_t *devices, int64_t n_pairs, const int64_t *offsets,
                                  const double *bvs1, const double *bvs2, const double *covs, const double *init_q,
                                  const double *init_t, const pnec_hip_pipeline_options *opt_in, double *out_q,
                                  double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count) {
  if (n_devices < 1 || !devices) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need at least one device");
  if (n_pairs < 0 || !offsets || !init_q || !init_t || !out_q || !out_t)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n_pairs == 0) return 0;
  pnec_hip_pipeline_options o;
  if (opt_in) o = *opt_in; else pnec_hip_default_pipeline_options(&o);
  if (offsets[0] != 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  // (a batch of empty pairs needs no covariances)
  if (!covs && !o.use_nec && offsets[n_pairs] > 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs is NULL for the PNEC chain");
  const int mode = covs || !o.use_nec ? PNEC_HIP_MODE_TARGET : PNEC_HIP_MODE_NEC;
  int64_t max_pair = 0;
  for (int64_t p = 0; p < n_pairs; ++p) max_pair = std::max(max_pair, offsets[p + 1] - offsets[p]);
  pnec_hip_multi *m = nullptr;
  if (int rc = pnec_hip_multi_create(n_devices, devices, mode, n_pairs, offsets[n_pairs], max_pair, &m)) return rc;
  int rc = pnec_hip_multi_fill(m, n_pairs, offsets, bvs1, bvs2, covs, nullptr);
  if (!rc) rc = pnec_hip_multi_solve_pipeline(m, init_q, init_t, &o, out_q, out_t, out_inlier_mask, out_inlier_count);
  const std::string msg = rc ? g_last_error : std::string();
  pnec_hip_multi_destroy(m);
  return rc ? fail(rc, msg) : 0;
}
