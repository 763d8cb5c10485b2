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
  const int mode = covs ? PNEC_HIP_MODE_TARGET : PNEC_HIP_MODE_NEC;
  if (!covs && !o.use_nec) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs is NULL for the PNEC chain");
  if (offsets[0] != 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  if (offsets[n_pairs] > 0 && (!bvs1 || !bvs2)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bvs1/bvs2 is NULL");
  int count = 0;
  PNEC_HIP_TRY(hipGetDeviceCount(&count));
  for (int32_t d = 0; d < n_devices; ++d)
    if (devices[d] < 0 || devices[d] >= count) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "no such device in the list");
  std::vector<int64_t> bounds((size_t)n_devices + 1);
  if (int rc = pnec_hip_partition(n_pairs, offsets, n_devices, bounds.data())) return rc;
  std::vector<int> rcs((size_t)n_devices, 0);
  std::vector<std::string> msgs((size_t)n_devices);
  auto work = [&](int32_t d) {
    const int64_t a = bounds[(size_t)d], z = bounds[(size_t)d + 1], m = z - a;
    if (m <= 0) return;
    std::vector<int64_t> off((size_t)m + 1);
    for (int64_t p = 0; p <= m; ++p) off[(size_t)p] = offsets[a + p] - offsets[a];
    pnec_hip_problem *prob = nullptr;
    hipStream_t stream = nullptr;
    int rc = pnec_hip_problem_create(devices[d], mode, m, off.data(), &prob);
    if (!rc) {
      DeviceGuard guard(devices[d]);
      // a stream of the shard's own: shards on the SAME device (a device listed twice) overlap instead of queueing
      // behind each other on the null stream
      if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) stream = nullptr;
      const int64_t c0 = offsets[a];
      if (off[(size_t)m] > 0)
        rc = pnec_hip_problem_fill(prob, 0, m, bvs1 + 3 * c0, bvs2 + 3 * c0, covs ? covs + 9 * c0 : nullptr, nullptr,
                                   PNEC_HIP_MEM_HOST, stream);
      if (!rc) {
        pnec_hip_pipeline_options po = o;
        po.first_pair_id = o.first_pair_id + a;   // the shard draws as the pairs it holds, not as pairs 0 .. m-1
        rc = pnec_hip_solve_pipeline(prob, init_q + 4 * a, init_t + 3 * a, &po, out_q + 4 * a, out_t + 3 * a,
                                     out_inlier_mask ? out_inlier_mask + c0 : nullptr,
                                     out_inlier_count ? out_inlier_count + a : nullptr, PNEC_HIP_MEM_HOST, stream);
      }
      if (rc) msgs[(size_t)d] = g_last_error;
      if (stream) { (void)hipStreamSynchronize(stream); }
    } else {
      msgs[(size_t)d] = g_last_error;
    }
    if (prob) pnec_hip_problem_destroy(prob);
    if (stream) {
      DeviceGuard guard(devices[d]);
      (void)hipStreamDestroy(stream);
    }
    rcs[(size_t)d] = rc;
  };
  std::vector<std::thread> threads;
  for (int32_t d = 1; d < n_devices; ++d) threads.emplace_back(work, d);
  work(0);  // (the calling thread takes the first shard)
  for (auto &t : threads) t.join();
  for (int32_t d = 0; d < n_devices; ++d)
    if (rcs[(size_t)d]) return fail(rcs[(size_t)d], "device " + std::to_string(devices[d]) + " (shard " + std::to_string(d) + "): " + msgs[(size_t)d]);
  return 0;
}
