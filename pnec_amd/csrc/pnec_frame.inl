// pnec_frame.inl -- part of pnec_capi.hip (inside extern "C"): the per-frame handle of the WHOLE chain.
//
// The reference's odometry calls PNEC::Solve once per frame pair (Frame2Frame::PNECAlign,
// src/rel_pose_estimation/frame2frame.cc:122-141 -> pnec.cc:77-124).  A batch object per call -- allocate,
// three blocking uploads, pack, the chain, blocking downloads, drain, free -- cost more host time than the
// chain's kernels take (and, once the chain forked a side stream per batch, milliseconds).  A frame handle owns
// everything a frame needs, once:
//   * a staging block in PINNED, device-mapped host memory: the caller's reference-layout arrays go in by
//     memcpy, the ingest kernel reads them over PCIe (zero-copy), the chain's last kernels write pose, inlier
//     mask and inlier count straight back into it -- no hipMemcpy in either direction;
//   * a capacity-shaped batch of one pair (pnec_hip_problem_create_capacity) that is re-shaped per frame on the
//     host side only: the ingest kernel writes the device-side index arrays (count, offsets) itself;
//   * the batch's cached scratch, InlierExtraction target and side stream (they live as long as the batch);
//   * one HIP stream; a frame is: memcpy in, ingest launch, pnec_hip_solve_pipeline in DEVICE space (the same
//     launches, hence the same bits, as the batch call), one stream synchronisation, memcpy out.
struct pnec_hip_frame {
  int device = 0;
  int64_t max_corr = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  char *h_base = nullptr, *d_base = nullptr;
  size_t o_q0 = 0, o_t0 = 0, o_b1 = 0, o_b2 = 0, o_cv = 0, o_oq = 0, o_ot = 0, o_cnt = 0, o_mask = 0;
  pnec_hip_problem *prob = nullptr;
  int64_t loaded = -1;  // correspondences of the frame currently in the batch (-1: none)
};

namespace {
// pack_kernel for the handle's single pair, with the pair's size as an argument: also writes the device-side
// index arrays of the (re-shaped) batch.  Same arithmetic as pack_kernel (symmetric part of the column-major
// 3x3, zeros in the padding), so the planes are bit for bit what pnec_hip_problem_fill would have produced.
__global__ __launch_bounds__(256) void frame_ingest_kernel(double *__restrict__ data, int64_t *__restrict__ block_offset,
                                                           int64_t *__restrict__ offsets, int32_t *__restrict__ count,
                                                           int n, const double *__restrict__ bvs1,
                                                           const double *__restrict__ bvs2,
                                                           const double *__restrict__ covs /* may be null: zeros */) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    block_offset[0] = 0;
    offsets[0] = 0;
    offsets[1] = n;
    count[0] = n;
  }
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += gridDim.x * blockDim.x) {
    const bool in = i < n;
    const int64_t j = in ? i : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      data[(int64_t)c * stride + i] = in ? bvs1[3 * j + c] : 0.0;
      data[(int64_t)(3 + c) * stride + i] = in ? bvs2[3 * j + c] : 0.0;
    }
    const bool cin = in && covs != nullptr;
    const double *C = covs + (cin ? 9 * j : 0);
    data[(int64_t)6 * stride + i] = cin ? C[0] : 0.0;
    data[(int64_t)7 * stride + i] = cin ? 0.5 * (C[1] + C[3]) : 0.0;
    data[(int64_t)8 * stride + i] = cin ? 0.5 * (C[2] + C[6]) : 0.0;
    data[(int64_t)9 * stride + i] = cin ? C[4] : 0.0;
    data[(int64_t)10 * stride + i] = cin ? 0.5 * (C[5] + C[7]) : 0.0;
    data[(int64_t)11 * stride + i] = cin ? C[8] : 0.0;
  }
}
}  // namespace

int pnec_hip_frame_create(int device, int64_t max_corr, void *stream_, pnec_hip_frame **out) {
  if (!out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (max_corr < 1 || max_corr > (int64_t)1 << 28) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need 1 <= max_corr <= 2^28");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed (no such device?)");
  pnec_hip_frame *f = new (std::nothrow) pnec_hip_frame();
  if (!f) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
  f->device = device;
  f->max_corr = max_corr;
  auto bail = [&](int rc) {
    const std::string msg = g_last_error;
    pnec_hip_frame_destroy(f);
    g_last_error = msg;
    return rc;
  };
  if (stream_) {
    f->stream = (hipStream_t)stream_;
    f->own_stream = false;
  } else {
    const hipError_t e = pool_stream_get(&f->stream);
    if (e != hipSuccess) return bail(fail_hip(e, "hipStreamCreate"));
  }
  const size_t M = (size_t)max_corr;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 63) / 64 * 64; return at; };
  f->o_q0 = take(sizeof(double) * 4);
  f->o_t0 = take(sizeof(double) * 3);
  f->o_oq = take(sizeof(double) * 4);
  f->o_ot = take(sizeof(double) * 3);
  f->o_cnt = take(sizeof(int32_t));
  f->o_b1 = take(sizeof(double) * 3 * M);
  f->o_b2 = take(sizeof(double) * 3 * M);
  f->o_cv = take(sizeof(double) * 9 * M);
  f->o_mask = take(M);
  void *h = nullptr, *d = nullptr;
  hipError_t e = hipHostMalloc(&h, (o + 4095) / 4096 * 4096, hipHostMallocMapped | hipHostMallocCoherent);
  if (e != hipSuccess) return bail(fail_hip(e, "hipHostMalloc(frame staging)"));
  f->h_base = (char *)h;
  e = hipHostGetDevicePointer(&d, h, 0);
  if (e != hipSuccess) return bail(fail_hip(e, "hipHostGetDevicePointer"));
  f->d_base = (char *)d;
  if (int rc = pnec_hip_problem_create_capacity(device, PNEC_HIP_MODE_TARGET, 1, max_corr, &f->prob)) return bail(rc);
  *out = f;
  return 0;
}

int pnec_hip_frame_destroy(pnec_hip_frame *f) {
  if (!f) return 0;
  DeviceGuard guard(f->device);
  bool drained = true;
  if (f->stream) drained = hipStreamSynchronize(f->stream) == hipSuccess;
  if (f->prob) pnec_hip_problem_destroy(f->prob);
  if (f->h_base) (void)hipHostFree(f->h_base);
  if (f->stream && f->own_stream) {
    if (drained) pool_stream_put(f->stream, f->device); else (void)hipStreamDestroy(f->stream);
  }
  delete f;
  return 0;
}

int64_t pnec_hip_frame_capacity(const pnec_hip_frame *f) { return f ? f->max_corr : 0; }

int pnec_hip_frame_load(pnec_hip_frame *f, int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                        pnec_hip_problem **problem) {
  if (!f) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "frame is NULL");
  if (problem) *problem = nullptr;
  if (n < 0 || n > f->max_corr) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "more correspondences than the handle was created for");
  if (n > 0 && (!bvs1 || !bvs2)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bvs1/bvs2 is NULL");
  DeviceGuard guard(f->device);
  // The ingest kernel reads the staging block over PCIe when it EXECUTES, and this call only enqueues it: a second
  // load -- or a load after DEVICE-space stage calls the caller queued on the handle's stream -- must not overwrite
  // bearings the GPU is still reading.  pnec_hip_frame_solve leaves the stream drained, so this costs a query there.
  {
    const hipError_t es = hipStreamSynchronize(f->stream);
    if (es != hipSuccess) return fail_hip(es, "pnec_hip_frame_load: work queued earlier on the handle's stream failed");
  }
  f->loaded = -1;
  char *h = f->h_base, *d = f->d_base;
  if (n > 0) {
    std::memcpy(h + f->o_b1, bvs1, sizeof(double) * 3 * (size_t)n);
    std::memcpy(h + f->o_b2, bvs2, sizeof(double) * 3 * (size_t)n);
    if (covs) std::memcpy(h + f->o_cv, covs, sizeof(double) * 9 * (size_t)n);
  }
  const int64_t offsets[2] = {0, n};
  if (int rc = problem_reshape_impl(f->prob, 1, offsets, f->stream, /*upload*/ false)) return rc;
  pnec_hip_problem *p = f->prob;
  const int stride = (int)((n + kWave - 1) & ~(int64_t)(kWave - 1));
  const unsigned blocks = (unsigned)std::max(1, std::min(64, (stride + 255) / 256));
  hipLaunchKernelGGL(frame_ingest_kernel, dim3(blocks), dim3(256), 0, f->stream, p->d_data, p->d_block_offset, p->d_offsets,
                     p->d_count, (int)n, (const double *)(d + f->o_b1), (const double *)(d + f->o_b2),
                     covs ? (const double *)(d + f->o_cv) : (const double *)nullptr);
  PNEC_HIP_TRY(hipGetLastError());
  f->loaded = n;
  if (problem) *problem = p;
  return 0;
}

void *pnec_hip_frame_stream(const pnec_hip_frame *f) { return f ? (void *)f->stream : nullptr; }

int pnec_hip_frame_solve(pnec_hip_frame *f, int64_t n, const double *bvs1, const double *bvs2, const double *covs,
                         const double *init_q, const double *init_t, const pnec_hip_pipeline_options *opt_in,
                         double *out_q, double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count) {
  if (!f || !init_q || !init_t || !out_q || !out_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  pnec_hip_pipeline_options o;
  if (opt_in) o = *opt_in; else pnec_hip_default_pipeline_options(&o);
  if (!o.use_nec && n > 0 && !covs) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs is NULL for the PNEC chain");
  if (int rc = pnec_hip_frame_load(f, n, bvs1, bvs2, covs, nullptr)) return rc;
  DeviceGuard guard(f->device);
  char *h = f->h_base, *d = f->d_base;
  std::memcpy(h + f->o_q0, init_q, sizeof(double) * 4);
  std::memcpy(h + f->o_t0, init_t, sizeof(double) * 3);
  // without RANSAC there are no inliers to report (inliers.clear(), pnec.cc:273-278): zeros, written here
  uint8_t *d_mask = o.use_ransac ? (uint8_t *)(d + f->o_mask) : nullptr;
  int32_t *d_cnt = o.use_ransac ? (int32_t *)(d + f->o_cnt) : nullptr;
  if (!o.use_ransac) {
    std::memset(h + f->o_cnt, 0, sizeof(int32_t));
    if (n > 0) std::memset(h + f->o_mask, 0, (size_t)n);
  }
  int rc = pnec_hip_solve_pipeline(f->prob, (const double *)(d + f->o_q0), (const double *)(d + f->o_t0), &o,
                                   (double *)(d + f->o_oq), (double *)(d + f->o_ot), d_mask, d_cnt, PNEC_HIP_MEM_DEVICE,
                                   f->stream);
  const hipError_t e = hipStreamSynchronize(f->stream);  // also after a failed launch: nothing may still be running
  if (rc) return rc;
  if (e != hipSuccess) return fail_hip(e, "PNEC::Solve chain");
  std::memcpy(out_q, h + f->o_oq, sizeof(double) * 4);
  std::memcpy(out_t, h + f->o_ot, sizeof(double) * 3);
  if (out_inlier_count) std::memcpy(out_inlier_count, h + f->o_cnt, sizeof(int32_t));
  if (out_inlier_mask && n > 0) std::memcpy(out_inlier_mask, h + f->o_mask, (size_t)n);
  return 0;
}
