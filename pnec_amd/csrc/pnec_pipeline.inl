// pnec_pipeline.inl -- part of pnec_capi.hip (inside extern "C"): PNEC::Solve's whole chain for a batch,
// device-resident from the first stage to the last (src/rel_pose_estimation/pnec.cc:77-124).
//
//   Eigensolver (+ RANSAC)  ->  InlierExtraction  ->  WeightedEigensolver + SCF  ->  CeresSolver
//   (or NECCeresSolver on the inliers when use_nec)
//
// Every stage is a launch on the caller's stream reading the previous stage's output in HBM; the inlier
// batch is compacted on the device into a cached batch of the same capacity, its sizes never visit the
// host.  HOST-space calls add one upload of the start poses in front and one download + wait at the end.

void pnec_hip_default_pipeline_options(pnec_hip_pipeline_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->use_ransac = 1;               // Options::use_ransac_            (pnec_config.h:57)
  o->use_nec = 0;                  // Options::use_nec_               (:47)
  o->use_ceres = 1;                // Options::use_ceres_             (:54)
  o->weighted_iterations = 10;     // Options::weighted_iterations_   (:52)
  o->max_ransac_iterations = 5000; // Options::max_ransac_iterations_ (:58)
  o->ransac_sample_size = 10;      // Options::ransac_sample_size_    (:59)
  o->regularization = 1.0e-13;     // Options::regularization_        (:51)
  o->ransac_threshold = 1.0e-6;    // pnec.cc:248
  o->ransac_seed = 1;
  pnec_hip_default_options(&o->solver);  // PNEC::CeresSolver default-constructs its optimiser (pnec.cc:355)
  o->eigensolver_scheme = PNEC_HIP_ES_NEWTON;
}

// this batch's bearings as a NEC-family batch: the first six planes of every block are f1 | f2 whatever
// the family, so the view shares every buffer and only differs in what the kernels are told to read
static pnec_hip_problem *nec_view_of(pnec_hip_problem *p) {
  if (p->mode == PNEC_HIP_MODE_NEC) return p;
  if (!p->nec_view) {
    pnec_hip_problem *v = new (std::nothrow) pnec_hip_problem();
    if (!v) return nullptr;
    v->owns_data = false;
    p->nec_view = v;
  }
  pnec_hip_problem *v = p->nec_view;
  v->device = p->device;
  v->mode = PNEC_HIP_MODE_NEC;
  v->nc = 6;
  v->n_pairs = p->n_pairs;
  v->n_corr = p->n_corr;
  v->n_max = p->n_max;
  v->data_doubles = p->data_doubles;
  v->d_data = p->d_data;
  v->d_block_offset = p->d_block_offset;
  v->d_offsets = p->d_offsets;
  v->d_count = p->d_count;
  if (v->host_counts != p->host_counts) {  // new sizes: the geometry buckets must be rebuilt
    v->host_counts = p->host_counts;
    v->buckets.clear();
    if (v->d_bucket_pairs) (void)dev_free(v->d_bucket_pairs);
    v->d_bucket_pairs = nullptr;
  }
  v->offsets = p->offsets;
  return v;
}

// Pairs [a, z) of `p` as a batch of their own: no data, the index arrays are slices of p's (block offsets and AoS
// offsets are absolute positions, so a slice is a valid array of the same kind); scratch, streams and views of its own.
// `v` is re-pointed on every call (p may have been re-shaped); its allocations stay.
static void chunk_view_repoint(pnec_hip_problem *v, pnec_hip_problem *p, int64_t a, int64_t z, bool lazy_bounds) {
  v->owns_data = false;
  v->device = p->device;
  v->mode = p->mode;
  v->nc = p->nc;
  v->n_pairs = z - a;
  v->data_doubles = p->data_doubles;
  v->d_data = p->d_data;
  v->d_block_offset = p->d_block_offset + a;
  v->d_offsets = p->d_offsets + a;
  v->d_count = p->d_count + a;
  std::vector<int32_t> hc(p->host_counts.begin() + a, p->host_counts.begin() + z);
  if (hc != v->host_counts) {   // new sizes: the geometry buckets must be rebuilt
    v->host_counts.swap(hc);
    v->buckets.clear();
    if (v->d_bucket_pairs) (void)dev_free(v->d_bucket_pairs);
    v->d_bucket_pairs = nullptr;
  }
  v->offsets.resize((size_t)(z - a) + 1);
  v->n_max = 0;
  for (int64_t i = a; i <= z; ++i) v->offsets[(size_t)(i - a)] = p->offsets[(size_t)i] - p->offsets[(size_t)a];
  for (int32_t c : v->host_counts) v->n_max = std::max(v->n_max, c);
  v->n_corr = v->offsets[(size_t)(z - a)];
  v->lazy = lazy_bounds;   // (an InlierExtraction target: the sizes above are the source's, i.e. upper bounds)
}

// where one chain run keeps its intermediate poses (slices of the caller batch's staging block)
struct PipelineScratch {
  double *es_q, *es_t, *w_q, *w_t;
};

// The chain on the batch `p` (a whole batch or a chunk view of one), everything in DEVICE space on `stream`.
// d_mask: the mask array of the batch p's AoS offsets index (for a chunk view: the parent's).  sv_given: the
// InlierExtraction target to compact into (a chunk view of the parent's), or null: p's own cached one is used.
static int pipeline_on(pnec_hip_problem *p, const double *d_iq, const double *d_it, const pnec_hip_pipeline_options &o,
                       double *d_oq, double *d_ot, uint8_t *d_mask, int32_t *d_cnt, bool want_count_zeros,
                       const PipelineScratch &sc, hipStream_t stream, pnec_hip_problem *sv_given) {
  const int64_t P = p->n_pairs;
  if (P == 0) return 0;
  if (int rc = ensure_front(p)) return rc;
  double *es_q = sc.es_q, *es_t = sc.es_t, *w_q = sc.w_q, *w_t = sc.w_t;
  hipError_t e = hipSuccess;
  pnec_hip_problem *stage = p;  // the batch the later stages run on (the inliers under RANSAC)
  // ---- ES_solution = Eigensolver(bvs1, bvs2, initial_pose, inliers); InlierExtraction
  if (o.use_ransac) {
    // the eigensolver on the inliers (latency-bound: sixteen pairs per wavefront, one wavefront per SIMD) runs
    // beside whatever follows that only needs the masks; both join before the weighted stage
    // (a handful of pairs -- the per-frame handle's one -- stay on one stream: the fork and the join through events
    // cost ~15 us, more than the two small launches take one after the other)
    const bool fork = P >= 1024;
    if (fork)
      if (int rc = ensure_side_streams(p, 1)) return rc;
    // InlierExtraction -- when a later stage works on the inliers.  With use_nec and no refinement (what the
    // reference's odometry forces, frame2frame.cc:127-128) or with neither weighted iterations nor refinement the chain
    // ends at the eigensolver's pose: the inlier mask and count are the outputs, and the copy would feed nothing.
    // It is FUSED into the RANSAC stage (round 4): the wavefront that has just scored a pair's best model over all its
    // correspondences compacts the kept ones into the target batch while the planes are still in the caches -- no
    // select launch, no second trip of the mask through HBM.
    const bool inliers_used = o.use_ceres || (!o.use_nec && o.weighted_iterations > 1);
    pnec_hip_problem *sv = nullptr;
    if (inliers_used) {
      sv = sv_given;
      if (!sv) {
        // (the cached InlierExtraction target has the capacity of the source; a re-shaped source is re-synced into it
        // by select_prepare, by layout generation)
        if (!p->sel_view || p->sel_view->cap_doubles < p->data_doubles || p->sel_view->cap_pairs < P) {
          if (p->sel_view) pnec_hip_problem_destroy(p->sel_view);
          p->sel_view = nullptr;
          if (int rc = alloc_like(p, stream, &p->sel_view)) return rc;
        }
        sv = p->sel_view;
        if (int rc = select_prepare(p, stream, sv)) return rc;
      }
    }
    if (int rc = ensure_order_hint(p)) return rc;
    e = launch_ransac_eigensolver(p->d_data, p->d_block_offset, p->d_offsets, p->d_count, P, d_iq, o.ransac_seed,
                                  (unsigned long long)o.first_pair_id, o.max_ransac_iterations, o.ransac_sample_size, o.ransac_threshold, es_q, es_t,
                                  d_mask, d_cnt, p->order_hint ? p->d_hint_its : nullptr, p->d_front, p->d_front_i, stream, fork ? p->side_streams[0] : nullptr,
                                  fork ? p->fork_event : nullptr, fork ? p->side_done[0] : nullptr, p->nc,
                                  sv ? sv->d_data : nullptr, sv ? sv->d_block_offset : nullptr, sv ? sv->d_count : nullptr,
                                  sv && !sv_given && P == 1 ? sv->d_offsets : nullptr,
                                  p->order_hint && p->order_pairs == P ? p->d_order : nullptr, o.eigensolver_scheme,
                                  o.ransac_flags);
    if (e == hipSuccess && p->order_hint) {   // the next call's launch order from this call's counts (pnec_hip_problem_launch_order_hint)
      e = launch_ransac_order(p->d_hint_its, P, p->d_order, stream);
      p->order_pairs = P;
    }
    if (e != hipSuccess) return fail_hip(e, "ransac_eigensolver_kernel");
    if (sv) {
      if (!sv_given)
        if (int rc = select_finish(p, stream, sv)) return rc;  // (the AoS offsets of the kept correspondences: one scan)
      stage = sv;
    }
    if (fork) PNEC_HIP_TRY(hipStreamWaitEvent(stream, p->side_done[0], 0));  // es_q / es_t are there from here on
  } else {
    e = launch_nec_eigensolver(p->d_data, p->d_block_offset, p->d_count, P, d_iq, es_q, es_t, nullptr, p->d_front,
                               p->d_front_i, stream, o.eigensolver_scheme);
    if (e != hipSuccess) return fail_hip(e, "nec_eigensolver_kernel");
    if (want_count_zeros) PNEC_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * P, stream));  // inliers.clear()
  }
  const double *res_q = es_q, *res_t = es_t;
  if (o.use_nec) {
    if (o.use_ceres) {  // NECCeresSolver(in_bvs1, in_bvs2, ES_solution)
      pnec_hip_problem *nv = nec_view_of(stage);
      if (!nv) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
      if (int rc = pnec_hip_solve(nv, es_q, es_t, 1, nullptr, 0.0, &o.solver, d_oq, d_ot, nullptr, nullptr, nullptr,
                                  PNEC_HIP_MEM_DEVICE, stream))
        return rc;
      res_q = nullptr;
    }
  } else {
    const double *ci_q = d_iq, *ci_t = d_it;  // weighted_iterations_ == 0: ceres_init = initial_pose
    if (o.weighted_iterations > 1) {
      // (the front scratch of p is free again: the RANSAC / eigensolver stage that used it is ahead on the stream)
      e = launch_weighted_eigensolver(p->device, stage->d_data, stage->d_block_offset, stage->d_count, P, stage->n_max,
                                      es_q, es_t, o.regularization, o.weighted_iterations, w_q, w_t, nullptr, p->d_front,
                                      p->d_front_i, stream, o.eigensolver_scheme);
      if (e != hipSuccess) return fail_hip(e, "weighted_eigensolver_kernel");
      ci_q = w_q; ci_t = w_t;
    } else if (o.weighted_iterations == 1) {
      ci_q = es_q; ci_t = es_t;
    }
    if (o.use_ceres) {  // CeresSolver(in_bvs1, in_bvs2, in_proj_covs, ceres_init)
      if (int rc = pnec_hip_solve(stage, ci_q, ci_t, 1, nullptr, o.regularization, &o.solver, d_oq, d_ot, nullptr,
                                  nullptr, nullptr, PNEC_HIP_MEM_DEVICE, stream))
        return rc;
      res_q = nullptr;
    } else {
      res_q = ci_q; res_t = ci_t;
    }
  }
  if (res_q) {  // the chain ended before the refinement: hand the last stage's pose out
    PNEC_HIP_TRY(hipMemcpyAsync(d_oq, res_q, sizeof(double) * 4 * P, hipMemcpyDeviceToDevice, stream));
    PNEC_HIP_TRY(hipMemcpyAsync(d_ot, res_t, sizeof(double) * 3 * P, hipMemcpyDeviceToDevice, stream));
  }
  return 0;
}

// How many ranges of pairs a batch's chain runs as, side by side: ONE, unless PNEC_PIPELINE_CHUNKS says otherwise (A/B).
// The idea -- the chain's kernels end in tails of a few long pairs (a quarter of the RANSAC launch at 20 000 pairs); let
// another range's kernels fill them, as several CALLS in flight do for a caller who has several batches -- does not
// carry over: ranges of one batch start together and reach their tails together, and each range's launches are a third
// of the size (1.6 generations of wavefronts instead of 5).  Measured, 20 000 pairs, same box, one call at a time /
// three calls in flight: 1 range 4.64 / 3.89 ms; 2 ranges 4.37 / 5.02; 3 ranges 5.00 / 5.69 (with eight hardware queues,
// GPU_MAX_HW_QUEUES=8: 4.56 / 3.54, 4.47 / 3.94, 4.53 / 4.68).  Results are bit for bit those of one range either way
// (RANSAC draws by global pair index); the whole -m gpu suite passes with three.
static int pipeline_chunks(int64_t P, bool use_ransac) {
  static const int forced = [] {
    const char *ev = std::getenv("PNEC_PIPELINE_CHUNKS");
    return ev && *ev ? std::atoi(ev) : 0;
  }();
  (void)use_ransac;
  if (forced >= 1) return (int)std::min<int64_t>(forced, std::max<int64_t>(P, 1));
  return 1;
}

int pnec_hip_solve_pipeline(pnec_hip_problem *p, const double *init_q, const double *init_t,
                            const pnec_hip_pipeline_options *opt_in, double *out_q, double *out_t,
                            uint8_t *out_inlier_mask, int32_t *out_inlier_count, int space, void *stream_) {
  if (!p || !init_q || !init_t || !out_q || !out_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  pnec_hip_pipeline_options o;
  if (opt_in) o = *opt_in; else pnec_hip_default_pipeline_options(&o);
  if (o.first_pair_id < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "first_pair_id < 0");
  if (o.weighted_iterations < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "weighted_iterations < 0");
  if (o.eigensolver_scheme < PNEC_HIP_ES_NEWTON || o.eigensolver_scheme > PNEC_HIP_ES_LM)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "eigensolver_scheme: 0 (Newton), 1 (descent) or 2 (LM)");
  if (o.eigensolver_scheme == PNEC_HIP_ES_DESCENT && !o.use_nec && o.weighted_iterations - 1 > kEsMaxRounds)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "eigensolver scheme 1 (descent) holds at most 16 weighted_iterations");
  if (!o.use_nec && p->mode != PNEC_HIP_MODE_TARGET)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "the PNEC chain needs a TARGET-mode problem (bearings + frame-2 covariances)");
  if (o.use_ransac && (o.max_ransac_iterations < 0 || o.ransac_sample_size < 1))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad RANSAC parameters");
  if (o.ransac_flags & ~PNEC_HIP_RANSAC_CHAINED_STARTS) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown RANSAC flag");
  if (o.use_ransac && o.ransac_sample_size > PNEC_HIP_MAX_RANSAC_SAMPLE)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "ransac sample_size > 16 is not built");
  const int64_t P = p->n_pairs;
  if (P == 0) return 0;
  if (space == PNEC_HIP_MEM_HOST && out_inlier_mask)
    if (int rc = materialize(p)) return rc;  // a host mask is sized by the exact total
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t M = std::max<int64_t>(p->n_corr, 1);

  // device scratch of the chain: [in_q 4P | in_t 3P | es_q 4P | es_t 3P | w_q 4P | w_t 3P | o_q 4P | o_t 3P], ints [cnt P | its P]
  if (int rc = ensure_stage(p, 28 * P, 2 * P)) return rc;
  double *w = p->d_stage;
  double *in_q = w; w += 4 * P;
  double *in_t = w; w += 3 * P;
  double *es_q = w; w += 4 * P;
  double *es_t = w; w += 3 * P;
  double *w_q = w;  w += 4 * P;
  double *w_t = w;  w += 3 * P;
  double *o_q = w;  w += 4 * P;
  double *o_t = w;
  int32_t *cnt = p->d_stage_i;
  const double *d_iq = init_q, *d_it = init_t;
  double *d_oq = out_q, *d_ot = out_t;
  uint8_t *d_mask = out_inlier_mask;
  int32_t *d_cnt = out_inlier_count ? out_inlier_count : cnt;
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(hipMemcpyAsync(in_q, init_q, sizeof(double) * 4 * P, hipMemcpyHostToDevice, stream));
    PNEC_HIP_TRY(hipMemcpyAsync(in_t, init_t, sizeof(double) * 3 * P, hipMemcpyHostToDevice, stream));
    d_iq = in_q; d_it = in_t; d_oq = o_q; d_ot = o_t; d_cnt = cnt; d_mask = nullptr;
  }
  if (o.use_ransac && !d_mask) {
    if (p->mask_bytes < M) {  // (a re-shaped batch may have grown)
      if (p->d_mask) (void)dev_free(p->d_mask);
      p->d_mask = nullptr;
      p->mask_bytes = 0;
      const int64_t want = std::max<int64_t>(M, p->cap_doubles / std::max(p->nc, 1));
      PNEC_HIP_TRY(dev_alloc(&p->d_mask, (size_t)want));
      p->mask_bytes = want;
    }
    d_mask = p->d_mask;
  }
  const int K = pipeline_chunks(P, o.use_ransac != 0);
  if (K <= 1) {
    const PipelineScratch sc{es_q, es_t, w_q, w_t};
    if (int rc = pipeline_on(p, d_iq, d_it, o, d_oq, d_ot, d_mask, d_cnt, out_inlier_count != nullptr, sc, stream, nullptr))
      return rc;
  } else {
    // ---- K contiguous ranges of the pairs, each the whole chain on its own stream (forked from the caller's and
    // joined into it); RANSAC draws by global pair index, so the results are those of the one-range run, bit for bit
    const bool inliers_used = o.use_ransac && (o.use_ceres || (!o.use_nec && o.weighted_iterations > 1));
    if (inliers_used) {
      if (!p->sel_view || p->sel_view->cap_doubles < p->data_doubles || p->sel_view->cap_pairs < P) {
        if (p->sel_view) pnec_hip_problem_destroy(p->sel_view);
        p->sel_view = nullptr;
        if (int rc = alloc_like(p, stream, &p->sel_view)) return rc;
      }
      if (int rc = select_prepare(p, stream, p->sel_view)) return rc;
    }
    while ((int)p->chunk_views.size() < K) {
      pnec_hip_problem *v = new (std::nothrow) pnec_hip_problem(), *sv = new (std::nothrow) pnec_hip_problem();
      hipStream_t st = nullptr;
      hipEvent_t ev = nullptr;
      hipError_t e = (v && sv) ? pool_stream_get(&st) : hipErrorOutOfMemory;
      if (e == hipSuccess) e = pool_event_get(&ev);
      if (e != hipSuccess) {
        delete v;
        delete sv;
        if (st) pool_stream_put(st, p->device);
        return fail_hip(e, "pipeline range (stream / event)");
      }
      v->owns_data = sv->owns_data = false;
      p->chunk_views.push_back(v);
      p->chunk_sel_views.push_back(sv);
      p->chunk_streams.push_back(st);
      p->chunk_done.push_back(ev);
    }
    if (!p->fork_event) PNEC_HIP_TRY(pool_event_get(&p->fork_event));
    PNEC_HIP_TRY(hipEventRecord(p->fork_event, stream));
    int rc_all = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t a = P * k / K, z = P * (k + 1) / K;
      pnec_hip_problem *v = p->chunk_views[(size_t)k], *sv = nullptr;
      chunk_view_repoint(v, p, a, z, p->lazy);
      if (inliers_used) {
        sv = p->chunk_sel_views[(size_t)k];
        chunk_view_repoint(sv, p->sel_view, a, z, true);
        sv->host_counts = v->host_counts;   // (the target's sizes are the source's bounds until somebody asks)
        sv->n_max = v->n_max;
      }
      hipStream_t cs = p->chunk_streams[(size_t)k];
      PNEC_HIP_TRY(hipStreamWaitEvent(cs, p->fork_event, 0));
      pnec_hip_pipeline_options ok = o;
      ok.first_pair_id = o.first_pair_id + a;
      const PipelineScratch sc{es_q + 4 * a, es_t + 3 * a, w_q + 4 * a, w_t + 3 * a};
      if (!rc_all)
        rc_all = pipeline_on(v, d_iq + 4 * a, d_it + 3 * a, ok, d_oq + 4 * a, d_ot + 3 * a, d_mask, d_cnt + a,
                             out_inlier_count != nullptr, sc, cs, sv);
      PNEC_HIP_TRY(hipEventRecord(p->chunk_done[(size_t)k], cs));
      PNEC_HIP_TRY(hipStreamWaitEvent(stream, p->chunk_done[(size_t)k], 0));
    }
    if (rc_all) return rc_all;
    if (inliers_used)
      if (int rc = select_finish(p, stream, p->sel_view)) return rc;   // (one scan over all ranges' counts)
  }
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(hipMemcpyAsync(out_q, d_oq, sizeof(double) * 4 * P, hipMemcpyDeviceToHost, stream));
    PNEC_HIP_TRY(hipMemcpyAsync(out_t, d_ot, sizeof(double) * 3 * P, hipMemcpyDeviceToHost, stream));
    if (o.use_ransac) {
      if (out_inlier_mask && p->n_corr > 0)
        PNEC_HIP_TRY(hipMemcpyAsync(out_inlier_mask, p->d_mask, (size_t)p->n_corr, hipMemcpyDeviceToHost, stream));
      if (out_inlier_count)
        PNEC_HIP_TRY(hipMemcpyAsync(out_inlier_count, cnt, sizeof(int32_t) * P, hipMemcpyDeviceToHost, stream));
    } else {
      if (out_inlier_mask && p->n_corr > 0) std::memset(out_inlier_mask, 0, (size_t)p->n_corr);
      if (out_inlier_count) std::memset(out_inlier_count, 0, sizeof(int32_t) * P);
    }
    PNEC_HIP_TRY(hipStreamSynchronize(stream));
  } else if (!o.use_ransac && out_inlier_mask && p->n_corr > 0) {
    PNEC_HIP_TRY(hipMemsetAsync(out_inlier_mask, 0, (size_t)p->n_corr, stream));
  }
  return 0;
}
