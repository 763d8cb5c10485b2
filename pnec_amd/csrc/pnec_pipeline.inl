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
  if (!o.use_nec && p->mode != PNEC_HIP_MODE_TARGET)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "the PNEC chain needs a TARGET-mode problem (bearings + frame-2 covariances)");
  if (o.use_ransac && (o.max_ransac_iterations < 0 || o.ransac_sample_size < 1))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad RANSAC parameters");
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
  if (int rc = ensure_front(p)) return rc;
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
  hipError_t e = hipSuccess;
  pnec_hip_problem *stage = p;  // the batch the later stages run on (the inliers under RANSAC)
  // ---- ES_solution = Eigensolver(bvs1, bvs2, initial_pose, inliers); InlierExtraction
  if (o.use_ransac) {
    if (!d_mask) {
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
    // the eigensolver on the inliers (latency-bound: sixteen pairs per wavefront, one wavefront per SIMD) runs
    // beside InlierExtraction (bandwidth-bound), which only needs the masks; both join before the weighted stage
    // (a handful of pairs -- the per-frame handle's one -- stay on one stream: the fork and the join through events
    // cost ~15 us, more than the two small launches take one after the other)
    const bool fork = P >= 1024;
    if (fork)
      if (int rc = ensure_side_streams(p, 1)) return rc;
    if (int rc = ensure_ransac_ws(p)) return rc;
    // InlierExtraction -- when a later stage works on the inliers.  With use_nec and no refinement (what the
    // reference's odometry forces, frame2frame.cc:127-128) or with neither weighted iterations nor refinement the chain
    // ends at the eigensolver's pose: the inlier mask and count are the outputs, and the copy would feed nothing.
    // It is FUSED into the RANSAC stage (round 4): the wavefront that has just scored a pair's best model over all its
    // correspondences compacts the kept ones into the target batch while the planes are still in the caches -- no
    // select launch, no second trip of the mask and the payload through HBM.
    const bool inliers_used = o.use_ceres || (!o.use_nec && o.weighted_iterations > 1);
    pnec_hip_problem *sv = nullptr;
    if (inliers_used) {
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
    e = launch_ransac_eigensolver(p->d_data, p->d_block_offset, p->d_offsets, p->d_count, P, d_iq, o.ransac_seed,
                                  (unsigned long long)o.first_pair_id, o.max_ransac_iterations, o.ransac_sample_size, o.ransac_threshold, es_q, es_t,
                                  d_mask, d_cnt, nullptr, p->d_front, p->d_front_i, stream, fork ? p->side_streams[0] : nullptr,
                                  fork ? p->fork_event : nullptr, fork ? p->side_done[0] : nullptr, p->nc,
                                  sv ? sv->d_data : nullptr, sv ? sv->d_block_offset : nullptr, sv ? sv->d_count : nullptr,
                                  sv && P == 1 ? sv->d_offsets : nullptr, p->d_ransac_ws);
    if (e != hipSuccess) return fail_hip(e, "ransac_eigensolver_kernel");
    if (sv) {
      if (int rc = select_finish(p, stream, sv)) return rc;  // (the AoS offsets of the kept correspondences: one scan)
      stage = sv;
    }
    if (fork) PNEC_HIP_TRY(hipStreamWaitEvent(stream, p->side_done[0], 0));  // es_q / es_t are there from here on
  } else {
    e = launch_nec_eigensolver(p->d_data, p->d_block_offset, p->d_count, P, d_iq, es_q, es_t, nullptr, p->d_front,
                               p->d_front_i, stream);
    if (e != hipSuccess) return fail_hip(e, "nec_eigensolver_kernel");
    if (out_inlier_count) PNEC_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t) * P, stream));  // inliers.clear()
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
                                      p->d_front_i, stream);
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
