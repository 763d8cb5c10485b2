// pnec_capi.hip -- the C ABI declared in include/pnec_hip.h: batch storage in HBM, ingest
// (reference AoS -> SoA planes), launch selection, and the small auxiliary kernels.
#include <hip/hip_runtime.h>

#include <thread>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <functional>
#include <mutex>
#include <vector>

#include "pnec_device.hpp"
#include "pnec_front_shared.hpp"
#include "pnec_solve_kernel.hpp"
#include "pnec_solve_group_kernel.hpp"

namespace pnec_hip {
// one translation unit per residual family (pnec_solve_<family>.hip)
hipError_t launch_solve_mode_0(int, int, int, bool, const SolveArgs &, hipStream_t);
hipError_t launch_solve_mode_1(int, int, int, bool, const SolveArgs &, hipStream_t);
hipError_t launch_solve_mode_2(int, int, int, bool, const SolveArgs &, hipStream_t);
hipError_t launch_solve_mode_3(int, int, int, bool, const SolveArgs &, hipStream_t);
// the multi-hypothesis form (pnec_solve_group_kernel.hpp): one block per (pair, group of hypotheses)
hipError_t launch_solve_group_mode_0(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_group_mode_1(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_group_mode_2(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_group_mode_3(int, int, int, const SolveArgs &, hipStream_t);

// pnec_stream_<family>.hip: the same kernels reading the reference's AoS arrays (streaming handle)
hipError_t launch_solve_aos_mode_0(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_aos_mode_1(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_aos_mode_2(int, int, int, const SolveArgs &, hipStream_t);
hipError_t launch_solve_aos_mode_3(int, int, int, const SolveArgs &, hipStream_t);

// pnec_frontend.hip
hipError_t launch_ransac_eigensolver(const double *, const int64_t *, const int64_t *, const int32_t *, int64_t,
                                     const double *, unsigned long long, unsigned long long, int, int, double, double *, double *,
                                     uint8_t *, int32_t *, int32_t *, double *, int32_t *, hipStream_t, hipStream_t,
                                     hipEvent_t, hipEvent_t, int, double *, const int64_t *, int32_t *, int64_t *, const int32_t *, int, int);
hipError_t launch_ransac_order(const int32_t *, int64_t, int32_t *, hipStream_t);
hipError_t frontend_work_counters(int, unsigned long long *, int *);
hipError_t launch_select(int, const double *, const int64_t *, const int64_t *, const int32_t *, const uint8_t *,
                         double *, const int64_t *, const int32_t *, int32_t *, int64_t *, int64_t, hipStream_t);
hipError_t launch_nec_eigensolver(const double *, const int64_t *, const int32_t *, int64_t, const double *,
                                  double *, double *, int32_t *, double *, int32_t *, hipStream_t, int);
hipError_t launch_weighted_eigensolver(int, const double *, const int64_t *, const int32_t *, int64_t, int,
                                       const double *, const double *, double, int, double *, double *,
                                       int32_t *, double *, int32_t *, hipStream_t, int);
hipError_t launch_frontend_selftest(double *, hipStream_t);
// scratch of the front stages, per pair (pnec_frontend.hip FrontScratch)
// (+ 3 kEsMaxRounds doubles and one int per pair: the weighted stage's chained minimisations under eigensolver schemes 1, 2;
//  + two ints per pair: the list of the RANSAC stage's second launch and its length)
}  // namespace pnec_hip

using namespace pnec_hip;

// ------------------------------------------------------------------------------------------
// HBM layout of a batch ("problem"):
//   data:  for pair p, a block of NC planes, each `stride_p = round_up(count_p, 64)` doubles:
//          f1x f1y f1z | f2x f2y f2z | cov xx xy xz yy yz zz | cov_host xx .. zz (SYM)
//          block_offset[p] = first double of the block; padding entries are 0.
//   A wavefront reading plane c touches 64 consecutive doubles (512 B) per load: fully coalesced.
struct pnec_hip_problem {
  int device = 0;
  int mode = 0;
  int nc = 0;
  int64_t n_pairs = 0;
  int64_t n_corr = 0;
  int32_t n_max = 0;
  int64_t data_doubles = 0;
  // A batch made with pnec_hip_problem_create_capacity is re-shaped in place (pnec_hip_problem_reshape): room for
  // cap_pairs pairs / cap_doubles doubles of planes, the index arrays laid out for cap_pairs.  0 = the shape it
  // was created with is all it can hold.  layout_gen counts the shapes it has had (views cache by it).
  int64_t cap_pairs = 0, cap_doubles = 0;
  uint64_t layout_gen = 0, view_src_gen = ~0ull;
  std::vector<int64_t> block_offset_host;  // reshape: the block layout of the current shape
  std::vector<int64_t> meta_host;      // reshape: the index arrays as uploaded (alive until the copy has run)
  hipEvent_t meta_uploaded = nullptr;  // reshape: recorded behind the upload
  std::vector<int64_t> offsets;       // host copy, [n_pairs+1]
  double *d_data = nullptr;           // SoA payload
  int64_t *d_block_offset = nullptr;  // [n_pairs]
  int64_t *d_offsets = nullptr;       // [n_pairs+1] AoS offsets (ingest only)
  int32_t *d_count = nullptr;         // [n_pairs]
  void *d_meta = nullptr;             // pnec_hip_problem_create: ONE block holding the three arrays above (one upload)
  // staging for host-space solves (grown on demand, reused)
  double *d_stage = nullptr;
  int64_t stage_doubles = 0;
  int32_t *d_stage_i = nullptr;
  int64_t stage_ints = 0;
  // scratch of the front stages (sums, starts and results of the batched eigenvalue minimisation)
  double *d_front = nullptr;
  int32_t *d_front_i = nullptr;
  int64_t front_pairs = 0;
  // which iteration the eigenvalue minimisations of the stage calls run (pnec_hip_problem_set_eigensolver_scheme)
  int es_scheme = 0;
  int ransac_flags = 0;   // PNEC_HIP_RANSAC_*: what pnec_hip_ransac_eigensolver on this batch runs with
  // capacity-shaped batches (filled again and again): the host-space fill's AoS staging, kept and grown on demand
  double *d_fill = nullptr;
  int64_t fill_doubles = 0;
  // launch-order hint of the RANSAC stage (pnec_hip_problem_launch_order_hint): the last run's hypothesis counts and
  // the order made from them; order_pairs = the number of pairs d_order is a permutation of (0: none yet)
  bool order_hint = false;
  int32_t *d_hint_its = nullptr, *d_order = nullptr;
  int64_t hint_cap = 0, order_pairs = 0;
  // ragged batches: pairs grouped by the smallest launch geometry that holds them (built lazily)
  struct Bucket {
    int cpl, wpp, ldsk;
    bool resident;
    int64_t count;
    int64_t first;  // offset into d_bucket_pairs
  };
  std::vector<Bucket> buckets;
  int32_t *d_bucket_pairs = nullptr;
  // ragged batches: the launches of the geometries in use run side by side (fork / join around them), so that
  // the long tail of one (solves of up to 50 iterations) is filled by the others' wavefronts
  std::vector<hipStream_t> side_streams;
  std::vector<hipEvent_t> side_done;
  hipEvent_t fork_event = nullptr;
  std::vector<int32_t> host_counts;
  // A batch produced by InlierExtraction on the device (pnec_hip_problem_select, the pipeline): it keeps
  // the source's block layout (capacity) and its real pair sizes exist only in d_count until somebody
  // asks for host-side numbers.  While `lazy`, host_counts / n_max / n_corr / offsets hold the SOURCE's
  // values, i.e. upper bounds -- all the launch selection needs.
  bool lazy = false;
  hipStream_t lazy_stream = nullptr;   // the stream the device-side sizes were produced on
  bool owns_data = true;               // false: a re-typed view of another batch's buffers (NEC view of a TARGET batch)
  pnec_hip_problem *sel_view = nullptr;  // pipeline: cached InlierExtraction target (same capacity, reused)
  pnec_hip_problem *nec_view = nullptr;  // pipeline: this batch's bearings as a NEC-family batch (no copy)
  // pipeline, large batches: contiguous ranges of the pairs as batches of their own (views: no data of their own,
  // index arrays = slices of this batch's), each with its scratch and its stream, and the same for the InlierExtraction
  // target -- the chain runs on them side by side (pnec_pipeline.inl)
  std::vector<pnec_hip_problem *> chunk_views, chunk_sel_views;
  std::vector<hipStream_t> chunk_streams;
  std::vector<hipEvent_t> chunk_done;
  uint8_t *d_mask = nullptr;             // pipeline: inlier mask [n_corr]
  int64_t mask_bytes = 0;
};

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
int fail_hip(hipError_t e, const char *what) {
  return fail(PNEC_HIP_ERR_HIP_RUNTIME, std::string(what) + ": " + hipGetErrorString(e));
}
#define PNEC_HIP_TRY(expr)                               \
  do {                                                   \
    hipError_t e_ = (expr);                              \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);    \
  } while (0)

// ---- device memory with a small cache --------------------------------------------------------
// Batches come and go in pipelines (create -> select -> destroy every frame set, or one batch per frame
// when the odometry calls PNEC::Solve), and hipMalloc / hipFree cost tens of microseconds for a small
// buffer and up to hundreds of ms for GB-sized ones on some boxes.  Freed blocks are kept (per device, up
// to PNEC_HIP_CACHE_MB, default 16384, and kMaxCachedBlocks blocks) and handed out again to requests of
// [size/2, size] -- requests below 1 MiB are rounded up to a power of two (>= 4 KiB) so that the small
// arrays of same-shaped batches always match; pnec_hip_release_cache() returns them to the driver.  A
// block is only cached after the device has drained (what hipFree does implicitly), so a new owner never
// races an old kernel; a batch's destructor drains once for all of its blocks (dev_free_drained).
struct DevBlock {
  void *ptr;
  size_t bytes;
  int device;
};
constexpr size_t kRoundBelowBytes = 1u << 20;  // requests below this are rounded up to a power of two
constexpr size_t kMaxCachedBlocks = 4096;
std::mutex g_mem_mutex;
std::unordered_map<void *, DevBlock> g_live;  // every block handed out
std::vector<DevBlock> g_cache;               // free blocks kept for reuse
size_t g_cached_bytes = 0;
uint64_t g_n_hip_malloc = 0, g_n_cache_hit = 0;   // pnec_hip_alloc_counters

size_t cache_limit_bytes() {
  static const size_t limit = [] {
    const char *e = std::getenv("PNEC_HIP_CACHE_MB");
    return (size_t)(e && *e ? std::strtoull(e, nullptr, 10) : 16384ull) << 20;
  }();
  return limit;
}

void release_cache_locked(int device /* -1: all */) {
  for (size_t i = 0; i < g_cache.size();) {
    if (device < 0 || g_cache[i].device == device) {
      (void)hipFree(g_cache[i].ptr);
      g_cached_bytes -= g_cache[i].bytes;
      g_cache[i] = g_cache.back();
      g_cache.pop_back();
    } else {
      ++i;
    }
  }
}

template <typename T>
hipError_t dev_alloc(T **out, size_t bytes) {
  *out = nullptr;
  if (bytes < kRoundBelowBytes) {
    size_t r = 4096;
    while (r < bytes) r <<= 1;
    bytes = r;
  }
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_mem_mutex);
  size_t best = g_cache.size();
  for (size_t i = 0; i < g_cache.size(); ++i)
    if (g_cache[i].device == device && g_cache[i].bytes >= bytes && g_cache[i].bytes <= 2 * bytes &&
        (best == g_cache.size() || g_cache[i].bytes < g_cache[best].bytes))
      best = i;
  DevBlock b{nullptr, bytes, device};
  if (best != g_cache.size()) {
    b = g_cache[best];
    g_cached_bytes -= b.bytes;
    g_cache[best] = g_cache.back();
    g_cache.pop_back();
    ++g_n_cache_hit;
  } else {
    ++g_n_hip_malloc;
    e = hipMalloc(&b.ptr, bytes);
    if (e != hipSuccess) {  // out of memory: give the cache back and try once more
      (void)hipGetLastError();
      release_cache_locked(device);
      e = hipMalloc(&b.ptr, bytes);
      if (e != hipSuccess) return e;
    }
  }
  g_live[b.ptr] = b;
  *out = static_cast<T *>(b.ptr);
  return hipSuccess;
}

// drained: the caller has synchronised the block's device since the last work that touched it
hipError_t dev_free_impl(void *ptr, bool drained) {
  if (!ptr) return hipSuccess;
  std::lock_guard<std::mutex> lock(g_mem_mutex);
  auto it = g_live.find(ptr);
  if (it == g_live.end()) return hipFree(ptr);
  const DevBlock b = it->second;
  g_live.erase(it);
  if (g_cache.size() < kMaxCachedBlocks && g_cached_bytes + b.bytes <= cache_limit_bytes()) {
    hipError_t e = hipSuccess;
    if (!drained) {
      int prev = -1;
      (void)hipGetDevice(&prev);
      (void)hipSetDevice(b.device);
      e = hipDeviceSynchronize();
      if (prev >= 0) (void)hipSetDevice(prev);
    }
    if (e == hipSuccess) {
      g_cache.push_back(b);
      g_cached_bytes += b.bytes;
      return hipSuccess;
    }
  }
  return hipFree(ptr);
}
hipError_t dev_free(void *ptr) { return dev_free_impl(ptr, false); }
hipError_t dev_free_drained(void *ptr) { return dev_free_impl(ptr, true); }

// ---- side streams and events with a pool -----------------------------------------------------
// hipStreamCreate / hipStreamDestroy cost milliseconds on some boxes (measured: a batch per frame that forked one
// side stream spent 2.7 of its 3.1 ms creating and destroying it).  Streams and events a batch no longer needs
// go back to a per-device pool (the owner drains before it lets go, like the memory blocks) and are handed out
// again; pnec_hip_release_cache() destroys them.
struct PooledStream {
  hipStream_t st;
  int device;
};
struct PooledEvent {
  hipEvent_t ev;
  int device;
};
std::vector<PooledStream> g_stream_pool;
std::vector<PooledEvent> g_event_pool;
constexpr size_t kMaxPooledStreams = 64, kMaxPooledEvents = 256;

hipError_t pool_stream_get(hipStream_t *out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(g_mem_mutex);
    for (size_t i = 0; i < g_stream_pool.size(); ++i)
      if (g_stream_pool[i].device == device) {
        *out = g_stream_pool[i].st;
        g_stream_pool[i] = g_stream_pool.back();
        g_stream_pool.pop_back();
        return hipSuccess;
      }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void pool_stream_put(hipStream_t st, int device) {
  if (!st) return;
  {
    std::lock_guard<std::mutex> lock(g_mem_mutex);
    if (g_stream_pool.size() < kMaxPooledStreams) {
      g_stream_pool.push_back({st, device});
      return;
    }
  }
  (void)hipStreamDestroy(st);
}
hipError_t pool_event_get(hipEvent_t *out) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  {
    std::lock_guard<std::mutex> lock(g_mem_mutex);
    for (size_t i = 0; i < g_event_pool.size(); ++i)
      if (g_event_pool[i].device == device) {
        *out = g_event_pool[i].ev;
        g_event_pool[i] = g_event_pool.back();
        g_event_pool.pop_back();
        return hipSuccess;
      }
  }
  return hipEventCreateWithFlags(out, hipEventDisableTiming);
}
void pool_event_put(hipEvent_t ev, int device) {
  if (!ev) return;
  {
    std::lock_guard<std::mutex> lock(g_mem_mutex);
    if (g_event_pool.size() < kMaxPooledEvents) {
      g_event_pool.push_back({ev, device});
      return;
    }
  }
  (void)hipEventDestroy(ev);
}
void release_stream_pool_locked(int device /* -1: all */) {
  for (size_t i = 0; i < g_stream_pool.size();) {
    if (device < 0 || g_stream_pool[i].device == device) {
      (void)hipStreamDestroy(g_stream_pool[i].st);
      g_stream_pool[i] = g_stream_pool.back();
      g_stream_pool.pop_back();
    } else {
      ++i;
    }
  }
  for (size_t i = 0; i < g_event_pool.size();) {
    if (device < 0 || g_event_pool[i].device == device) {
      (void)hipEventDestroy(g_event_pool[i].ev);
      g_event_pool[i] = g_event_pool.back();
      g_event_pool.pop_back();
    } else {
      ++i;
    }
  }
}

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// ---- ingest: reference AoS (bvs 3, covs 9 column-major) -> SoA planes --------------------
template <int NC>
__global__ __launch_bounds__(256) void pack_kernel(double *__restrict__ data,
                                                   const int64_t *__restrict__ block_offset,
                                                   const int64_t *__restrict__ offsets,
                                                   const int32_t *__restrict__ count,
                                                   int64_t first_pair, int64_t n_pairs,
                                                   const double *__restrict__ bvs1,
                                                   const double *__restrict__ bvs2,
                                                   const double *__restrict__ covs,
                                                   const double *__restrict__ covs_host) {
  const int64_t src0 = offsets[first_pair];
  for (int64_t p = first_pair + blockIdx.y; p < first_pair + n_pairs; p += gridDim.y) {
    const int n = count[p];
    const int stride = (n + kWave - 1) & ~(kWave - 1);
    double *blk = data + block_offset[p];
    const int64_t src = offsets[p] - src0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += gridDim.x * blockDim.x) {
      const bool in = i < n;
      const int64_t j = src + i;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        blk[(int64_t)c * stride + i] = in ? bvs1[3 * j + c] : 0.0;
        blk[(int64_t)(3 + c) * stride + i] = in ? bvs2[3 * j + c] : 0.0;
      }
      if constexpr (NC >= 12) {
        // symmetric part of the column-major 3x3: (r,c) at 3*c + r
        const double *C = covs + 9 * j;
        blk[(int64_t)6 * stride + i] = in ? C[0] : 0.0;
        blk[(int64_t)7 * stride + i] = in ? 0.5 * (C[1] + C[3]) : 0.0;
        blk[(int64_t)8 * stride + i] = in ? 0.5 * (C[2] + C[6]) : 0.0;
        blk[(int64_t)9 * stride + i] = in ? C[4] : 0.0;
        blk[(int64_t)10 * stride + i] = in ? 0.5 * (C[5] + C[7]) : 0.0;
        blk[(int64_t)11 * stride + i] = in ? C[8] : 0.0;
      }
      if constexpr (NC >= 18) {
        const double *C = covs_host + 9 * j;
        blk[(int64_t)12 * stride + i] = in ? C[0] : 0.0;
        blk[(int64_t)13 * stride + i] = in ? 0.5 * (C[1] + C[3]) : 0.0;
        blk[(int64_t)14 * stride + i] = in ? 0.5 * (C[2] + C[6]) : 0.0;
        blk[(int64_t)15 * stride + i] = in ? C[4] : 0.0;
        blk[(int64_t)16 * stride + i] = in ? 0.5 * (C[5] + C[7]) : 0.0;
        blk[(int64_t)17 * stride + i] = in ? C[8] : 0.0;
      }
    }
  }
}

// ---- best hypothesis per pair ------------------------------------------------------------
__global__ void select_best_kernel(int64_t n_pairs, int n_hyp, const double *__restrict__ cost,
                                   int32_t *__restrict__ best) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int bi = 0;
  double bc = cost[p * n_hyp];
  for (int h = 1; h < n_hyp; ++h) {
    const double c = cost[p * n_hyp + h];
    // NaN never wins; first NaN-free minimum wins ties
    if (c < bc || (bc != bc && c == c)) {
      bc = c;
      bi = h;
    }
  }
  best[p] = bi;
}

// ---- pnec::common::CostFunction (common.cc:237-259), one wavefront per pair ---------------
__global__ __launch_bounds__(kWave) void cost_function_kernel(const double *__restrict__ data,
                                                              const int64_t *__restrict__ block_offset,
                                                              const int32_t *__restrict__ count,
                                                              const double *__restrict__ qs,
                                                              const double *__restrict__ ts,
                                                              double *__restrict__ out) {
  const int64_t p = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = count[p];
  const int stride = (n + kWave - 1) & ~(kWave - 1);
  const double *base = data + block_offset[p];
  double q[4] = {qs[4 * p], qs[4 * p + 1], qs[4 * p + 2], qs[4 * p + 3]};
  const double qn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] *= qn;
  double R[9];
  rot_from_quat(q, R);
  const double tx = ts[3 * p], ty = ts[3 * p + 1], tz = ts[3 * p + 2];
  double acc = 0.0;
  for (int i = lane; i < n; i += kWave) {
    double d[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) d[c] = base[(int64_t)c * stride + i];
    const double mx = ty * d[2] - tz * d[1], my = tz * d[0] - tx * d[2], mz = tx * d[1] - ty * d[0];
    const double gx = R[0] * mx + R[3] * my + R[6] * mz;
    const double gy = R[1] * mx + R[4] * my + R[7] * mz;
    const double gz = R[2] * mx + R[5] * my + R[8] * mz;
    const double nn = d[3] * gx + d[4] * gy + d[5] * gz;
    const double sgx = d[6] * gx + d[7] * gy + d[8] * gz;
    const double sgy = d[7] * gx + d[9] * gy + d[10] * gz;
    const double sgz = d[8] * gx + d[10] * gy + d[11] * gz;
    acc += nn * nn / (gx * sgx + gy * sgy + gz * sgz);
  }
  acc = wave_allreduce_sum(acc);
  if (lane == 0) out[p] = acc / (double)n;
}

// ---- covariance propagation: pnec::common::UnscentedTransform + Unproject -----------------
// (src/common/common.cc:460-525; 5 sigma points, kappa-weighted).  All matrices column-major like
// Eigen.  camera_model: 0 omnidirectional, 1 pinhole.  One function shared by the stand-alone kernel and
// the fused keypoint ingest, compiled WITHOUT floating-point contraction so that both produce the same
// bits whatever code surrounds the call (the ingest test compares them bitwise).
//   m:  the image point (x, y, 1) [or (x, y, f) with K_inv = I];  c0, c1: the two columns added to /
//   subtracted from it (columns of the covariance's Cholesky factor).
__device__ __forceinline__ void unscented_core(const double (&m)[3], const double (&c0)[3], const double (&c1)[3],
                                               const double (&K)[9], double kappa, int camera_model,
                                               double (&bearing)[3], double (&S)[9]) {
#pragma clang fp contract(off)
  const double w0 = kappa / (2.0 + kappa), wi = 0.5 / (2.0 + kappa);
  double tp[5][3], mean[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const double sg = (p == 0) ? 0.0 : (p <= 2 ? 1.0 : -1.0);
    const double *col = (p == 1 || p == 3) ? c0 : c1;
    const double x = m[0] + sg * col[0], y = m[1] + sg * col[1], z = m[2] + sg * col[2];
    double tx = x, ty = y, tz = z;
    if (camera_model != 0) {
      tx = K[0] * x + K[3] * y + K[6] * z;
      ty = K[1] * x + K[4] * y + K[7] * z;
      tz = K[2] * x + K[5] * y + K[8] * z;
    }
    const double nn = 1.0 / sqrt(tx * tx + ty * ty + tz * tz);
    tp[p][0] = tx * nn; tp[p][1] = ty * nn; tp[p][2] = tz * nn;
    const double w = (p == 0) ? w0 : wi;
#pragma unroll
    for (int k = 0; k < 3; ++k) mean[k] += w * tp[p][k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) S[k] = 0.0;
#pragma unroll
  for (int p = 0; p < 5; ++p) {
    const double w = (p == 0) ? w0 : wi;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) S[3 * c + r] += w * (tp[p][r] - mean[r]) * (tp[p][c] - mean[c]);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) bearing[k] = tp[0][k];  // normalised (K^-1) mu = Unproject
}
// the pinhole branch's sigma-point offsets: columns of the lower Cholesky factor of the image-plane
// covariance [[a, b], [b, d]] (common.cc:488-489)
__device__ __forceinline__ void pinhole_columns(double a, double b, double d, double (&c0)[3], double (&c1)[3]) {
#pragma clang fp contract(off)
  const double l00 = sqrt(a), l10 = b / l00, l11 = sqrt(d - l10 * l10);
  c0[0] = l00; c0[1] = l10; c0[2] = 0.0;
  c1[0] = 0.0; c1[1] = l11; c1[2] = 0.0;
}

__global__ __launch_bounds__(256) void unscented_kernel(int64_t n, const double *__restrict__ mu,
                                                        const double *__restrict__ covs,
                                                        const double *__restrict__ K_inv_, double kappa,
                                                        int camera_model, double *__restrict__ out_bvs,
                                                        double *__restrict__ out_covs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double K[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) K[k] = K_inv_[k];
  const double m[3] = {mu[3 * i], mu[3 * i + 1], mu[3 * i + 2]};
  const double *C9 = covs + 9 * i;
  double c0[3], c1[3];  // the two columns added to / subtracted from mu
  if (camera_model == 0) {
    // rotation taking (0,0,1) to the bearing (RotationBetweenPoints, common.cc:118-124)
    const double nm = fast_rsqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    const double vx = m[0] * nm, vy = m[1] * nm, vz = m[2] * nm;
    const double cx = -vy, cy = vx;  // (0,0,1) x v = (-vy, vx, 0)
    double R[9];                     // column-major
    const double f = 1.0 / (1.0 + vz);
    // K = skew(c) = [[0,0,cy],[0,0,-cx],[-cy,cx,0]];  R = I + K + K^2 f
    R[0] = 1.0 - cy * cy * f; R[3] = cx * cy * f;       R[6] = cy;
    R[1] = cx * cy * f;       R[4] = 1.0 - cx * cx * f; R[7] = -cx;
    R[2] = -cy;               R[5] = cx;                R[8] = 1.0 - (cx * cx + cy * cy) * f;
    // local = (R' cov R) top-left 2x2
    double T[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[3 * c + r] = C9[r] * R[3 * c] + C9[3 + r] * R[3 * c + 1] + C9[6 + r] * R[3 * c + 2];
    const double a = R[0] * T[0] + R[1] * T[1] + R[2] * T[2];
    const double b = R[3] * T[0] + R[4] * T[1] + R[5] * T[2];
    const double d = R[3] * T[3] + R[4] * T[4] + R[5] * T[5];
    const double l00 = sqrt(a), l10 = b / l00, l11 = sqrt(d - l10 * l10);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      c0[r] = R[r] * l00 + R[3 + r] * l10;
      c1[r] = R[3 + r] * l11;
    }
  } else {
    pinhole_columns(C9[0], C9[1], C9[4], c0, c1);
  }
  double bearing[3], S[9];
  unscented_core(m, c0, c1, K, kappa, camera_model, bearing, S);
#pragma unroll
  for (int k = 0; k < 9; ++k) out_covs[9 * i + k] = S[k];
  if (out_bvs) {
#pragma unroll
    for (int k = 0; k < 3; ++k) out_bvs[3 * i + k] = bearing[k];
  }
}

// ---- fused keypoint ingest: KeyPoint::Unproject (src/frames/keypoints.cc:49-62) for both frames'
// keypoints -- bearing = normalised K^-1 (u, v, 1), covariance = UnscentedTransform of the 2x2 image
// covariance, kappa = 1, pinhole -- written straight into the batch's SoA planes.  Per correspondence the
// device reads 56 B (two pixel positions, one symmetric 2x2) instead of the 120 B of ready-made bearings
// + 3x3 covariance, and the AoS covariances never exist in HBM.  Same bits as unscented_kernel followed
// by pack_kernel (both call unscented_core; the 3x3 it returns is exactly symmetric, so pack_kernel's
// symmetrisation is the identity on it).
template <int NC>
__global__ __launch_bounds__(256) void ingest_keypoints_kernel(double *__restrict__ data,
                                                               const int64_t *__restrict__ block_offset,
                                                               const int64_t *__restrict__ offsets,
                                                               const int32_t *__restrict__ count, int64_t first_pair,
                                                               int64_t n_pairs, const double *__restrict__ pts1,
                                                               const double *__restrict__ pts2,
                                                               const double *__restrict__ cov2,
                                                               const double *__restrict__ cov1,
                                                               const double *__restrict__ K_inv_, double kappa) {
  double K[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) K[k] = K_inv_[k];
  const int64_t src0 = offsets[first_pair];
  const double zero3[3] = {0.0, 0.0, 0.0};
  for (int64_t p = first_pair + blockIdx.y; p < first_pair + n_pairs; p += gridDim.y) {
    const int n = count[p];
    const int stride = (n + kWave - 1) & ~(kWave - 1);
    double *blk = data + block_offset[p];
    const int64_t src = offsets[p] - src0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < stride; i += gridDim.x * blockDim.x) {
      const bool in = i < n;
      const int64_t j = in ? src + i : src;  // a valid address for the padding lanes
      double b1[3], b2[3], S2[9], S1[9];
      {
        const double m[3] = {pts2[2 * j], pts2[2 * j + 1], 1.0};
        double c0[3], c1[3];
        if constexpr (NC >= 12) pinhole_columns(cov2[3 * j], cov2[3 * j + 1], cov2[3 * j + 2], c0, c1);
        else { c0[0] = c0[1] = c0[2] = c1[0] = c1[1] = c1[2] = 0.0; }
        unscented_core(m, c0, c1, K, kappa, 1, b2, S2);
      }
      {
        const double m[3] = {pts1[2 * j], pts1[2 * j + 1], 1.0};
        if constexpr (NC >= 18) {
          double c0[3], c1[3];
          pinhole_columns(cov1[3 * j], cov1[3 * j + 1], cov1[3 * j + 2], c0, c1);
          unscented_core(m, c0, c1, K, kappa, 1, b1, S1);
        } else {
          unscented_core(m, zero3, zero3, K, kappa, 1, b1, S1);  // only the bearing is kept
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        blk[(int64_t)c * stride + i] = in ? b1[c] : 0.0;
        blk[(int64_t)(3 + c) * stride + i] = in ? b2[c] : 0.0;
      }
      if constexpr (NC >= 12) {
        blk[(int64_t)6 * stride + i] = in ? S2[0] : 0.0;
        blk[(int64_t)7 * stride + i] = in ? 0.5 * (S2[1] + S2[3]) : 0.0;
        blk[(int64_t)8 * stride + i] = in ? 0.5 * (S2[2] + S2[6]) : 0.0;
        blk[(int64_t)9 * stride + i] = in ? S2[4] : 0.0;
        blk[(int64_t)10 * stride + i] = in ? 0.5 * (S2[5] + S2[7]) : 0.0;
        blk[(int64_t)11 * stride + i] = in ? S2[8] : 0.0;
      }
      if constexpr (NC >= 18) {
        blk[(int64_t)12 * stride + i] = in ? S1[0] : 0.0;
        blk[(int64_t)13 * stride + i] = in ? 0.5 * (S1[1] + S1[3]) : 0.0;
        blk[(int64_t)14 * stride + i] = in ? 0.5 * (S1[2] + S1[6]) : 0.0;
        blk[(int64_t)15 * stride + i] = in ? S1[4] : 0.0;
        blk[(int64_t)16 * stride + i] = in ? 0.5 * (S1[5] + S1[7]) : 0.0;
        blk[(int64_t)17 * stride + i] = in ? S1[8] : 0.0;
      }
    }
  }
}

// ---- inliers per pair from a correspondence mask ----------------------------------------------
__global__ __launch_bounds__(kWave) void mask_count_kernel(const uint8_t *__restrict__ mask,
                                                           const int64_t *__restrict__ offsets,
                                                           const int32_t *__restrict__ count,
                                                           int32_t *__restrict__ out,
                                                           int64_t *__restrict__ single_offsets /* null, or the
                                                           new batch's offsets when it has ONE pair: the scan of one
                                                           count is the count (a launch less per frame) */) {
  const int64_t p = blockIdx.x;
  const int n = count[p];
  int c = 0;
  for (int i = threadIdx.x; i < n; i += kWave) c += mask[offsets[p] + i] != 0;
  c = (int)wave_allreduce_sum((double)c);
  if (threadIdx.x == 0) {
    out[p] = c;
    if (single_offsets) {
      single_offsets[0] = 0;
      single_offsets[1] = c;
    }
  }
}

// exclusive prefix sum of the pair sizes -> AoS offsets [n+1] (one workgroup; n is at most a few 1e5).  Segments of
// 32 x 1024 pairs: each thread takes up to 32 consecutive sizes, ALL of them requested before the first is used; the 1024
// partial sums are scanned by shuffles inside the sixteen wavefronts + one scan of their totals.  (Until round 6: one
// dependent load after the other per thread, twice, and thread 0 walking the 1024 partial sums alone -- ~40 us on the
// chain's critical path in every call.)
__global__ __launch_bounds__(1024) void offsets_scan_kernel(const int32_t *__restrict__ count,
                                                            int64_t *__restrict__ offsets, int64_t n) {
  constexpr int kPer = 32;
  __shared__ long long wave_tot[17];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  long long carry = 0;
  for (int64_t seg = 0; seg < n; seg += (int64_t)kPer * 1024) {
    const int64_t m = std::min<int64_t>(n - seg, (int64_t)kPer * 1024);
    const int64_t per = (m + 1023) / 1024, a = seg + std::min<int64_t>(m, per * t), b = seg + std::min<int64_t>(m, per * (t + 1));
    int32_t c[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) c[k] = (a + k < b) ? count[a + k] : 0;
    long long sacc = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) sacc += c[k];
    long long inc = sacc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    __syncthreads();   // (the previous segment's totals have been read)
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    if (wave == 0) {
      const long long w = lane < 16 ? wave_tot[lane] : 0ll;
      long long winc = w;
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) {
        const long long o = __shfl_up(winc, d, 64);
        if (lane >= d) winc += o;
      }
      if (lane < 16) wave_tot[lane] = winc - w;
      if (lane == 15) wave_tot[16] = winc;
    }
    __syncthreads();
    long long run = carry + wave_tot[wave] + inc - sacc;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (a + k < b) offsets[a + k] = run;
      run += c[k];
    }
    carry += wave_tot[16];
  }
  if (t == 0) offsets[n] = carry;
}

// ---- device self-test kernels (cross-lane reduction, 5x5 solve) ---------------------------
__global__ void selftest_kernel(double *out) {
  const int lane = threadIdx.x;
  // sum of (lane+1)^2 over 64 lanes = 89440; every lane must hold it
  const double v = wave_allreduce_sum((double)((lane + 1) * (lane + 1)));
  out[lane] = v;
  if (lane == 0) {
    // A = M M' + I for a fixed M; solve A y = b and report the residual norm
    double P[15], b[5], y[5];
    double M[5][5];
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) M[i][j] = sin(1.0 + i * 1.7 + j * 0.9);
    for (int i = 0; i < 5; ++i)
      for (int j = i; j < 5; ++j) {
        double sacc = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < 5; ++k) sacc += M[i][k] * M[j][k];
        P[tri(i, j)] = sacc;
      }
    for (int i = 0; i < 5; ++i) b[i] = 1.0 + i;
    const bool ok = chol_solve5(P, b, y);
    double res = 0.0;
    for (int i = 0; i < 5; ++i) {
      double sacc = -b[i];
      for (int j = 0; j < 5; ++j) sacc += P[sym(i, j)] * y[j];
      res += sacc * sacc;
    }
    out[64] = ok ? sqrt(res) : -1.0;
    out[65] = fast_rsqrt(2.0) - 0.70710678118654752440;
    out[66] = fast_rcp(3.0) - 0.33333333333333333333;
  }
  {
    // the same system solved across lanes (gj_solve5_rows: lane i of every 16-lane row owns row i) with a damped
    // diagonal, against chol_solve5 on the damped matrix; and the 64-bit row broadcast on its own
    double P[15], bb[5], y[5];
    double M[5][5];
    for (int i = 0; i < 5; ++i)
      for (int j = 0; j < 5; ++j) M[i][j] = sin(1.0 + i * 1.7 + j * 0.9);
    for (int i = 0; i < 5; ++i)
      for (int j = i; j < 5; ++j) {
        double sacc = (i == j) ? 1.0 : 0.0;
        for (int k = 0; k < 5; ++k) sacc += M[i][k] * M[j][k];
        P[tri(i, j)] = sacc;
      }
    for (int i = 0; i < 5; ++i) bb[i] = 1.0 + i;
    const int li = lane & 15;
    double A[5], rhs = 0.0, damp = 0.0;
    for (int k = 0; k < 5; ++k) A[k] = 0.0;
    for (int i = 0; i < 5; ++i)
      if (li == i) {
        for (int k = 0; k < 5; ++k) A[k] = P[sym(i, k)];
        rhs = bb[i];
        damp = 0.25 * (i + 1);
      }
    for (int i = 0; i < 5; ++i) P[tri(i, i)] += 0.25 * (i + 1);
    const bool ok_rows = gj_solve5_rows(A, damp, rhs, li);
    const bool ok_chol = chol_solve5(P, bb, y);
    double dev = 0.0;
    for (int i = 0; i < 5; ++i)
      if (li == i) dev = fabs(rhs - y[i]) / fabs(y[i]);
    out[216 + lane] = (ok_rows && ok_chol && li < 5) ? dev : ((ok_rows && ok_chol) ? 0.0 : -1.0);
    out[280 + lane] = bcast_row<11>(1000.0 + lane);  // must be 1011 + 16 * (lane / 16)
  }
  // 21-way swap-halving reduction: acc[j] = (lane+1)(j+1) + j  ->  2080 (j+1) + 64 j
  double acc[kNumAcc], sums[kNumAcc];
  for (int j = 0; j < kNumAcc; ++j) acc[j] = (double)((lane + 1) * (j + 1) + j);
  wave_reduce21(acc, sums);
  if (lane == 0)
    for (int j = 0; j < kNumAcc; ++j) out[67 + j] = sums[j];
  // lean acos / atan2 against libm: a grid over the circle and over [-1, 1] with dense ends
  {
    double worst_a = 0.0;
    for (int k = 0; k < 64; ++k) {
      const int i = lane * 64 + k;
      const double ang = -3.14159 + 6.28318 * i / 4095.0;
      const double yy = sin(ang) * (1.0 + (i % 7)), xx = cos(ang) * (1.0 + (i % 7));
      worst_a = fmax(worst_a, fabs(atan2_lean(yy, xx) - atan2(yy, xx)));
      const double u = (double)i / 4095.0;
      const double c = (i & 1) ? 1.0 - u * u * u * u : -1.0 + u * u * u * u;   // clusters at +-1
      const double ref = acos(c);
      worst_a = fmax(worst_a, fabs(acos_lean(c) - ref) / fmax(ref, 1e-300));
    }
    out[152 + lane] = worst_a;
  }
  // bounded sincos against libm over [-40, 40]
  double worst = 0.0;
  for (int k = 0; k < 64; ++k) {
    const double x = -40.0 + 80.0 * (lane * 64 + k) / 4095.0;
    double s1, c1, s2, c2;
    sincos_bounded(x, s1, c1);
    sincos(x, &s2, &c2);
    worst = fmax(worst, fmax(fabs(s1 - s2), fabs(c1 - c2)));
  }
  out[88 + lane] = worst;
}

// ---- launch geometry --------------------------------------------------------------------
struct Geometry {
  int cpl, wpp, ldsk;
  bool resident;
};


bool geometry_exists(int mode, int cpl, int wpp, int ldsk) {
#define PNEC_GEOMETRY_MATCH(CPL, WPP, LDSK) \
  if (cpl == CPL && wpp == WPP && ldsk == LDSK) return geometry_ok(mode, cpl, wpp, ldsk);
  PNEC_FOR_EACH_GEOMETRY(PNEC_GEOMETRY_MATCH)
#undef PNEC_GEOMETRY_MATCH
  return false;
}

// The auto-tuner's ladder for a residual family (smallest capacity first).
// planes = true: the batch path, which reads the SoA planes in HBM and has the tail form (12, 1, 3) for pairs of
// 513..768 -- one wavefront, 512 correspondences resident, the tail re-read from L2 every pass -- whose results are
// bit for bit those of (8, 2, 3); the AoS-source kernels of the streaming handle (planes = false) are not built for it
// and run such pairs on (8, 2, 3), to the same bits.
int geometry_ladder(int mode, const int (**order)[3], bool planes = true) {
  static const int order12t[][3] = {{1, 1, 0}, {2, 1, 0}, {4, 1, 0}, {8, 1, 3}, {12, 1, 3},
                                    {8, 2, 3}, {8, 4, 3}, {8, 8, 3}};
  static const int order6t[][3] = {{1, 1, 0}, {2, 1, 0}, {4, 1, 0}, {8, 1, 0}, {12, 1, 3},
                                   {8, 2, 3}, {8, 4, 3}, {8, 8, 3}};
  if (planes && mode != PNEC_HIP_MODE_SYM) {
    *order = (mode == PNEC_HIP_MODE_NEC) ? order6t : order12t;
    return 8;
  }
  static const int order12[][3] = {{1, 1, 0}, {2, 1, 0}, {4, 1, 0}, {8, 1, 3},
                                   {8, 2, 3}, {8, 4, 3}, {8, 8, 3}};
  // 18-plane payload: 512 correspondences fit ONE wavefront at one wavefront per SIMD (288 payload
  // registers, AGPRs included) and that beats two wavefronts per solve: 19.0 vs 15.7 M solves/s
  static const int order18[][3] = {{1, 1, 0}, {2, 1, 0}, {4, 1, 0}, {8, 1, 0}, {4, 4, 0}, {4, 8, 0}};
  // 6-plane NEC payload: 8 correspondences per lane are 96 registers, so 512 fit one wavefront at two
  // per SIMD without the LDS slots: 41.7 vs 38.8 M solves/s
  static const int order6[][3] = {{1, 1, 0}, {2, 1, 0}, {4, 1, 0}, {8, 1, 0},
                                  {8, 2, 3}, {8, 4, 3}, {8, 8, 3}};
  if (mode == PNEC_HIP_MODE_SYM) { *order = order18; return 6; }
  *order = (mode == PNEC_HIP_MODE_NEC) ? order6 : order12;
  return 7;
}

// On-chip resident whenever the largest pair fits 64*CPL*WPP slots.  Preference: as few
// wavefronts per solve as possible (the serial part of an LM iteration is paid once per
// wavefront) at two wavefronts per SIMD; 12-plane payloads use the (8,W,3) family (5 of a
// lane's 8 correspondences in registers, 3 in LDS), the 18-plane SYM payload (8,1,0) up to 512
// correspondences and the (4,W,0) family beyond.
int choose_geometry(const pnec_hip_problem *p, const pnec_hip_options *opt, Geometry *g) {
  const int n = std::max<int32_t>(p->n_max, 1);
  if (opt && (opt->corr_per_lane > 0 || opt->waves_per_pair > 0)) {
    const int cpl = opt->corr_per_lane, wpp = opt->waves_per_pair ? opt->waves_per_pair : 1;
    const int ldsk = opt->lds_corr_per_lane;
    if (cpl < 0 || wpp < 0 || ldsk < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "negative launch tuning");
    if (cpl == 0) {  // streaming with the fixed block shape
      *g = {1, kStreamWaves, 0, false};
      return 0;
    }
    if (!geometry_exists(p->mode, cpl, wpp, ldsk))
      return fail(PNEC_HIP_ERR_UNSUPPORTED,
                  "launch geometry (corr_per_lane, waves_per_pair, lds_corr_per_lane) not built");
    if ((int64_t)kWave * cpl * wpp < n)
      return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "launch geometry too small for the largest pair");
    *g = {cpl, wpp, ldsk, true};
    return 0;
  }
  const int (*order)[3];
  const int count = geometry_ladder(p->mode, &order);
  for (int i = 0; i < count; ++i) {
    if ((int64_t)kWave * order[i][0] * order[i][1] >= n) {
      *g = {order[i][0], order[i][1], order[i][2], true};
      return 0;
    }
  }
  *g = {1, kStreamWaves, 0, false};
  return 0;
}

// Ragged batches: one launch per geometry actually needed, each over the pairs that fit it, so a
// few large pairs do not force every small pair into a many-wavefront geometry.
int ensure_buckets(pnec_hip_problem *p) {
  if (!p->buckets.empty() || p->n_pairs == 0) return 0;
  const int (*order)[3];
  const int count = geometry_ladder(p->mode, &order);
  std::vector<std::vector<int32_t>> lists((size_t)count + 1);
  for (int64_t i = 0; i < p->n_pairs; ++i) {
    const int n = std::max<int32_t>(p->host_counts[(size_t)i], 1);
    int b = count;  // streaming
    for (int k = 0; k < count; ++k)
      if ((int64_t)kWave * order[k][0] * order[k][1] >= n) {
        b = k;
        break;
      }
    lists[(size_t)b].push_back((int32_t)i);
  }
  std::vector<int32_t> flat;
  std::vector<pnec_hip_problem::Bucket> buckets;
  flat.reserve((size_t)p->n_pairs);
  for (int b = 0; b <= count; ++b) {
    if (lists[(size_t)b].empty()) continue;
    pnec_hip_problem::Bucket bk;
    if (b < count) {
      bk = {order[b][0], order[b][1], order[b][2], true, (int64_t)lists[(size_t)b].size(), (int64_t)flat.size()};
    } else {
      bk = {1, kStreamWaves, 0, false, (int64_t)lists[(size_t)b].size(), (int64_t)flat.size()};
    }
    buckets.push_back(bk);
    flat.insert(flat.end(), lists[(size_t)b].begin(), lists[(size_t)b].end());
  }
  if (buckets.size() == 1) {  // one geometry serves every pair: the launch indexes pairs directly, no table
    p->buckets = std::move(buckets);
    return 0;
  }
  int32_t *d_pairs = nullptr;
  PNEC_HIP_TRY(dev_alloc(&d_pairs, sizeof(int32_t) * flat.size()));
  const hipError_t e = hipMemcpy(d_pairs, flat.data(), sizeof(int32_t) * flat.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)dev_free(d_pairs);
    return fail_hip(e, "hipMemcpy(bucket pairs)");
  }
  // published only now: a failed attempt leaves the problem as it was (no buckets, no table)
  p->d_bucket_pairs = d_pairs;
  p->buckets = std::move(buckets);
  return 0;
}

int ensure_stage(pnec_hip_problem *p, int64_t doubles, int64_t ints) {
  if (doubles > p->stage_doubles) {
    if (p->d_stage) (void)dev_free(p->d_stage);
    p->d_stage = nullptr;
    p->stage_doubles = 0;
    PNEC_HIP_TRY(dev_alloc(&p->d_stage, sizeof(double) * doubles));
    p->stage_doubles = doubles;
  }
  if (ints > p->stage_ints) {
    if (p->d_stage_i) (void)dev_free(p->d_stage_i);
    p->d_stage_i = nullptr;
    p->stage_ints = 0;
    PNEC_HIP_TRY(dev_alloc(&p->d_stage_i, sizeof(int32_t) * ints));
    p->stage_ints = ints;
  }
  return 0;
}

// at least `n` side streams (+ their "done" events) and the fork event of the batch
int ensure_side_streams(pnec_hip_problem *p, size_t n) {
  while (p->side_streams.size() < n) {
    hipStream_t st = nullptr;
    hipEvent_t ev = nullptr;
    hipError_t e = pool_stream_get(&st);
    if (e == hipSuccess) e = pool_event_get(&ev);
    if (e != hipSuccess) {
      pool_stream_put(st, p->device);
      return fail_hip(e, "side stream");
    }
    p->side_streams.push_back(st);
    p->side_done.push_back(ev);
  }
  if (!p->fork_event) PNEC_HIP_TRY(pool_event_get(&p->fork_event));
  return 0;
}

// The launch-order hint's arrays (two int32 per pair), grown on demand.
int ensure_order_hint(pnec_hip_problem *p) {
  if (!p->order_hint || p->hint_cap >= p->n_pairs) return 0;
  if (p->d_hint_its) (void)dev_free(p->d_hint_its);
  if (p->d_order) (void)dev_free(p->d_order);
  p->d_hint_its = p->d_order = nullptr;
  p->hint_cap = p->order_pairs = 0;
  const int64_t want = std::max<int64_t>(p->n_pairs, p->cap_pairs);
  PNEC_HIP_TRY(dev_alloc(&p->d_hint_its, sizeof(int32_t) * (size_t)want));
  PNEC_HIP_TRY(dev_alloc(&p->d_order, sizeof(int32_t) * (size_t)want));
  p->hint_cap = want;
  return 0;
}

int ensure_front(pnec_hip_problem *p) {
  const int64_t P = std::max<int64_t>(p->n_pairs, 1);
  if (P > p->front_pairs) {
    if (p->d_front) (void)dev_free(p->d_front);
    if (p->d_front_i) (void)dev_free(p->d_front_i);
    p->d_front = nullptr;
    p->d_front_i = nullptr;
    p->front_pairs = 0;
    PNEC_HIP_TRY(dev_alloc(&p->d_front, sizeof(double) * (size_t)kFrontDoublesPerPair * P));
    PNEC_HIP_TRY(dev_alloc(&p->d_front_i, sizeof(int32_t) * ((size_t)kFrontIntsPerPair * P + kFrontCounterInts)));
    p->front_pairs = P;
  }
  return 0;
}

}  // namespace

// ==========================================================================================
extern "C" {

static int materialize(const pnec_hip_problem *cp);
static int solve_work_buffer(int device, unsigned long long **out);

int pnec_hip_abi_version(void) { return PNEC_HIP_ABI_VERSION; }

int pnec_hip_problem_launch_order_hint(pnec_hip_problem *p, int32_t enable) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL problem");
  p->order_hint = enable != 0;
  if (!p->order_hint) p->order_pairs = 0;
  return 0;
}

const char *pnec_hip_last_error(void) { return g_last_error.c_str(); }

int pnec_hip_device_count(int *count) {
  if (!count) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "count is NULL");
  *count = 0;
  PNEC_HIP_TRY(hipGetDeviceCount(count));
  return 0;
}

void pnec_hip_default_options(pnec_hip_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->check_convergence = 1;
  o->corr_per_lane = 0;
  o->waves_per_pair = 0;
  o->lds_corr_per_lane = 0;
  o->flags = 0;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
}

// the block layout of a shape: block_offset / count per pair, total doubles, the largest pair
static int shape_layout(int nc, int64_t n_pairs, const int64_t *offsets, std::vector<int64_t> &block_offset,
                        std::vector<int32_t> &count, int64_t *total_out, int32_t *n_max_out) {
  if (n_pairs < 0 || !offsets) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad n_pairs/offsets");
  if (offsets[0] != 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets[0] must be 0");
  block_offset.resize((size_t)n_pairs);
  count.resize((size_t)n_pairs);
  int64_t total = 0;
  int32_t n_max = 0;
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t n = offsets[p + 1] - offsets[p];
    if (n < 0 || n > (int64_t)1 << 30)
      return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing (pair sizes < 2^30)");
    const int64_t stride = (n + kWave - 1) & ~(int64_t)(kWave - 1);
    block_offset[(size_t)p] = total;
    count[(size_t)p] = (int32_t)n;
    total += stride * nc;
    n_max = std::max<int32_t>(n_max, (int32_t)n);
  }
  *total_out = total;
  *n_max_out = n_max;
  return 0;
}

// block_offset [C] | offsets [C+1] | count [C] (C = the pair capacity) in one device block, filled by one copy (a
// batch per frame pays every blocking copy in full: three of them were a tenth of the one-pair PNEC::Solve)
static int upload_meta(pnec_hip_problem *p, const std::vector<int64_t> &block_offset, const int64_t *offsets,
                       const std::vector<int32_t> &count, hipStream_t stream, bool blocking) {
  const int64_t n_pairs = (int64_t)count.size();
  const size_t C1 = (size_t)std::max<int64_t>(std::max(p->cap_pairs, n_pairs), 1);
  const size_t words = C1 + (C1 + 1);  // int64 entries
  // an asynchronous upload reads this buffer when the stream gets there: it lives in the batch, and the next
  // upload waits for the previous one before it overwrites it
  if (p->meta_uploaded) PNEC_HIP_TRY(hipEventSynchronize(p->meta_uploaded));
  std::vector<int64_t> &meta = p->meta_host;
  meta.assign(words + (C1 + 1) / 2, 0);
  std::copy(block_offset.begin(), block_offset.end(), meta.begin());
  std::copy(offsets, offsets + n_pairs + 1, meta.begin() + (ptrdiff_t)C1);
  std::memcpy(meta.data() + words, count.data(), sizeof(int32_t) * count.size());
  if (!p->d_meta) {
    int64_t *d_meta = nullptr;
    hipError_t e = dev_alloc(&d_meta, sizeof(int64_t) * meta.size());
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(meta)");
    p->d_meta = d_meta;
    p->d_block_offset = d_meta;
    p->d_offsets = d_meta + C1;
    p->d_count = reinterpret_cast<int32_t *>(d_meta + words);
  }
  // the used prefix of each array is what changes; for the small capacities of per-frame handles one copy
  // of the whole block is cheaper than three
  hipError_t e;
  if (blocking) {
    e = hipMemcpy(p->d_meta, meta.data(), sizeof(int64_t) * meta.size(), hipMemcpyHostToDevice);
  } else if (C1 <= 4096) {
    e = hipMemcpyAsync(p->d_meta, meta.data(), sizeof(int64_t) * meta.size(), hipMemcpyHostToDevice, stream);
  } else {
    e = hipMemcpyAsync(p->d_block_offset, meta.data(), sizeof(int64_t) * (size_t)std::max<int64_t>(n_pairs, 1),
                       hipMemcpyHostToDevice, stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(p->d_offsets, meta.data() + C1, sizeof(int64_t) * (size_t)(n_pairs + 1), hipMemcpyHostToDevice,
                         stream);
    if (e == hipSuccess)
      e = hipMemcpyAsync(p->d_count, meta.data() + words, sizeof(int32_t) * (size_t)std::max<int64_t>(n_pairs, 1),
                         hipMemcpyHostToDevice, stream);
  }
  if (e != hipSuccess) return fail_hip(e, "hipMemcpy(meta)");
  if (!blocking) {
    if (!p->meta_uploaded) PNEC_HIP_TRY(pool_event_get(&p->meta_uploaded));
    PNEC_HIP_TRY(hipEventRecord(p->meta_uploaded, stream));
  }
  return 0;
}

static int problem_create_impl(int device, int mode, int64_t cap_pairs, int64_t cap_doubles, int64_t n_pairs,
                               const int64_t *offsets, pnec_hip_problem **out) {
  if (!out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (mode < PNEC_HIP_MODE_NEC || mode > PNEC_HIP_MODE_SYM)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown mode");
  std::vector<int64_t> block_offset;
  std::vector<int32_t> count;
  const int nc = num_components(mode);
  int64_t total = 0;
  int32_t n_max = 0;
  if (int rc = shape_layout(nc, n_pairs, offsets, block_offset, count, &total, &n_max)) return rc;
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed (no such device?)");
  pnec_hip_problem *p = new (std::nothrow) pnec_hip_problem();
  if (!p) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
  p->device = device;
  p->mode = mode;
  p->nc = nc;
  p->n_pairs = n_pairs;
  p->n_corr = offsets[n_pairs];
  p->n_max = n_max;
  p->host_counts = count;
  p->data_doubles = total;
  p->cap_pairs = cap_pairs;
  p->cap_doubles = cap_doubles;
  p->offsets.assign(offsets, offsets + n_pairs + 1);
  hipError_t e;
  // (+ kDataSlackDoubles: the weighted stage's 16-byte table loads may read 512 bytes past the last pair's last plane)
  if ((e = dev_alloc(&p->d_data, sizeof(double) * (std::max<int64_t>(std::max(total, cap_doubles), 1) + kDataSlackDoubles))) != hipSuccess) {
    pnec_hip_problem_destroy(p);
    return fail_hip(e, "hipMalloc(data)");
  }
  if (int rc = upload_meta(p, block_offset, offsets, count, nullptr, /*blocking*/ true)) {
    const std::string msg = g_last_error;
    pnec_hip_problem_destroy(p);
    g_last_error = msg;
    return rc;
  }
  *out = p;
  return 0;
}

int pnec_hip_problem_create(int device, int mode, int64_t n_pairs, const int64_t *offsets,
                            pnec_hip_problem **out) {
  return problem_create_impl(device, mode, 0, 0, n_pairs, offsets, out);
}

int pnec_hip_problem_create_capacity(int device, int mode, int64_t max_pairs, int64_t max_corr,
                                     pnec_hip_problem **out) {
  if (!out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (mode < PNEC_HIP_MODE_NEC || mode > PNEC_HIP_MODE_SYM) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown mode");
  if (max_pairs < 1 || max_corr < 0 || max_corr > (int64_t)1 << 40)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "need max_pairs >= 1, max_corr >= 0");
  // every pair's planes are padded to a multiple of 64 correspondences: at most 63 extra per pair
  const int64_t cap_doubles = (int64_t)num_components(mode) * ((max_corr + 63 * max_pairs + kWave - 1) & ~(int64_t)(kWave - 1));
  const int64_t zero = 0;
  return problem_create_impl(device, mode, max_pairs, cap_doubles, 0, &zero, out);
}

static int problem_reshape_impl(pnec_hip_problem *p, int64_t n_pairs, const int64_t *offsets, hipStream_t stream,
                                bool upload);
int pnec_hip_problem_reshape(pnec_hip_problem *p, int64_t n_pairs, const int64_t *offsets, void *stream_) {
  return problem_reshape_impl(p, n_pairs, offsets, (hipStream_t)stream_, true);
}
// upload = false: the caller's next kernel on the stream writes the device-side index arrays itself (the frame
// handle's ingest kernel does, for its single pair)
static int problem_reshape_impl(pnec_hip_problem *p, int64_t n_pairs, const int64_t *offsets, hipStream_t stream,
                                bool upload) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  // (d_meta: the index arrays in ONE block, which only pnec_hip_problem_create[_capacity] makes -- an InlierExtraction
  // target has capacity too, but its index arrays are separate allocations that upload_meta would leak and replace)
  if (p->cap_pairs <= 0 || !p->owns_data || !p->d_meta)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "not a capacity-shaped batch (pnec_hip_problem_create_capacity)");
  std::vector<int64_t> block_offset;
  std::vector<int32_t> count;
  int64_t total = 0;
  int32_t n_max = 0;
  if (int rc = shape_layout(p->nc, n_pairs, offsets, block_offset, count, &total, &n_max)) return rc;
  if (n_pairs > p->cap_pairs || total > p->cap_doubles)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "shape exceeds the batch's capacity");
  DeviceGuard guard(p->device);
  if (upload)
    if (int rc = upload_meta(p, block_offset, offsets, count, stream, /*blocking*/ false)) return rc;
  p->n_pairs = n_pairs;
  p->n_corr = offsets[n_pairs];
  p->n_max = n_max;
  p->host_counts = count;
  p->data_doubles = total;
  p->offsets.assign(offsets, offsets + n_pairs + 1);
  p->lazy = false;
  p->buckets.clear();
  if (p->d_bucket_pairs) (void)dev_free(p->d_bucket_pairs);  // (drains: only ragged multi-geometry shapes have one)
  p->d_bucket_pairs = nullptr;
  // views cache the block layout by generation: a new one only when it really changed (the frame handle's single
  // pair always starts at 0, so its InlierExtraction target never re-copies anything)
  if (block_offset != p->block_offset_host) {
    p->block_offset_host = block_offset;
    ++p->layout_gen;
  }
  return 0;
}

int pnec_hip_problem_destroy(pnec_hip_problem *p) {
  if (!p) return 0;
  DeviceGuard guard(p->device);
  // one drain for all of the batch's blocks (and its views'): nothing launched on them is still running
  // when they go back to the cache
  const bool drained = hipDeviceSynchronize() == hipSuccess;
  auto release = [&](void *ptr) { (void)dev_free_impl(ptr, drained); };
  for (pnec_hip_problem *v : p->chunk_views) pnec_hip_problem_destroy(v);
  for (pnec_hip_problem *v : p->chunk_sel_views) pnec_hip_problem_destroy(v);
  if (p->sel_view) pnec_hip_problem_destroy(p->sel_view);
  if (p->nec_view) pnec_hip_problem_destroy(p->nec_view);
  release(p->d_mask);
  if (p->owns_data) {
    release(p->d_data);
    if (p->d_meta) {
      release(p->d_meta);
    } else {
      release(p->d_block_offset);
      release(p->d_offsets);
      release(p->d_count);
    }
  }
  release(p->d_stage);
  release(p->d_stage_i);
  release(p->d_front);
  release(p->d_front_i);
  release(p->d_hint_its);
  release(p->d_order);
  // (the device has drained: nothing is pending on these, so the next owner starts clean)
  for (hipStream_t st : p->side_streams) {
    if (drained) pool_stream_put(st, p->device); else (void)hipStreamDestroy(st);
  }
  for (hipEvent_t ev : p->side_done) {
    if (drained) pool_event_put(ev, p->device); else (void)hipEventDestroy(ev);
  }
  for (hipStream_t st : p->chunk_streams) {
    if (drained) pool_stream_put(st, p->device); else (void)hipStreamDestroy(st);
  }
  for (hipEvent_t ev : p->chunk_done) {
    if (drained) pool_event_put(ev, p->device); else (void)hipEventDestroy(ev);
  }
  if (p->fork_event) {
    if (drained) pool_event_put(p->fork_event, p->device); else (void)hipEventDestroy(p->fork_event);
  }
  if (p->meta_uploaded) {
    if (drained) pool_event_put(p->meta_uploaded, p->device); else (void)hipEventDestroy(p->meta_uploaded);
  }
  release(p->d_bucket_pairs);
  release(p->d_fill);
  delete p;
  return 0;
}

int pnec_hip_problem_fill(pnec_hip_problem *p, int64_t first_pair, int64_t n_pairs,
                          const double *bvs1, const double *bvs2, const double *covs,
                          const double *covs_host, int space, void *stream_) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  if (first_pair < 0 || n_pairs < 0 || first_pair + n_pairs > p->n_pairs)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pair range out of bounds");
  if (n_pairs == 0) return 0;
  if (int rc = materialize(p)) return rc;
  const int64_t m = p->offsets[(size_t)(first_pair + n_pairs)] - p->offsets[(size_t)first_pair];
  if (m > 0 && (!bvs1 || !bvs2)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bvs1/bvs2 is NULL");
  if (m > 0 && p->nc >= 12 && !covs)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs is NULL for a PNEC-mode problem");
  if (m > 0 && p->nc >= 18 && !covs_host)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "covs_host is NULL for a SYM-mode problem");
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;

  const double *d_b1 = bvs1, *d_b2 = bvs2, *d_c = covs, *d_ch = covs_host;
  double *tmp = nullptr;
  bool persistent_stage = false;
  if (space == PNEC_HIP_MEM_HOST && m > 0) {
    const int64_t per = 6 + (p->nc >= 12 ? 9 : 0) + (p->nc >= 18 ? 9 : 0);
    // a batch that is re-filled keeps its staging (nothing allocated per call: the per-frame and streaming handles) --
    // up to 256 MB; beyond that the staging is ~1.25 x the payload (6 GB for 100k x 512) and is borrowed from the
    // library's buffer cache per call instead, so that it is shared by every batch on the device, not held by each
    if (p->cap_pairs > 0 && per * m * (int64_t)sizeof(double) <= (256ll << 20)) {
      if (per * m > p->fill_doubles) {
        if (p->d_fill) (void)dev_free(p->d_fill);
        p->d_fill = nullptr;
        p->fill_doubles = 0;
        PNEC_HIP_TRY(dev_alloc(&p->d_fill, sizeof(double) * per * m));
        p->fill_doubles = per * m;
      }
      persistent_stage = true;
      tmp = p->d_fill;
    } else {
      PNEC_HIP_TRY(dev_alloc(&tmp, sizeof(double) * per * m));
    }
    double *w = tmp;
    auto up = [&](const double *src, int64_t k, const double **dst) -> hipError_t {
      *dst = w;
      hipError_t e = hipMemcpyAsync(w, src, sizeof(double) * k * m, hipMemcpyHostToDevice, stream);
      w += k * m;
      return e;
    };
    hipError_t e = up(bvs1, 3, &d_b1);
    if (e == hipSuccess) e = up(bvs2, 3, &d_b2);
    if (e == hipSuccess && p->nc >= 12) e = up(covs, 9, &d_c);
    if (e == hipSuccess && p->nc >= 18) e = up(covs_host, 9, &d_ch);
    if (e != hipSuccess) {
      if (!persistent_stage) (void)dev_free(tmp);
      return fail_hip(e, "hipMemcpyAsync(H2D)");
    }
  } else if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST) {
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  }

  const int32_t n_max = std::max<int32_t>(p->n_max, 1);
  const dim3 block(256);
  const dim3 grid((unsigned)std::min<int64_t>((n_max + 255) / 256, 64),
                  (unsigned)std::min<int64_t>(n_pairs, 32768));
  switch (p->nc) {
    case 6:
      hipLaunchKernelGGL(pack_kernel<6>, grid, block, 0, stream, p->d_data, p->d_block_offset,
                         p->d_offsets, p->d_count, first_pair, n_pairs, d_b1, d_b2, d_c, d_ch);
      break;
    case 12:
      hipLaunchKernelGGL(pack_kernel<12>, grid, block, 0, stream, p->d_data, p->d_block_offset,
                         p->d_offsets, p->d_count, first_pair, n_pairs, d_b1, d_b2, d_c, d_ch);
      break;
    default:
      hipLaunchKernelGGL(pack_kernel<18>, grid, block, 0, stream, p->d_data, p->d_block_offset,
                         p->d_offsets, p->d_count, first_pair, n_pairs, d_b1, d_b2, d_c, d_ch);
      break;
  }
  hipError_t e = hipGetLastError();
  if (tmp) {
    if (e == hipSuccess) e = hipStreamSynchronize(stream);   // (the caller may reuse its arrays; the staging may be refilled)
    if (!persistent_stage) (void)dev_free(tmp);
  }
  if (e != hipSuccess) return fail_hip(e, "pack_kernel");
  return 0;
}

int pnec_hip_problem_fill_keypoints(pnec_hip_problem *p, int64_t first_pair, int64_t n_pairs, const double *pts1,
                                    const double *pts2, const double *cov2, const double *cov1, const double *K_inv,
                                    double kappa, int camera_model, int space, void *stream_) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  if (first_pair < 0 || n_pairs < 0 || first_pair + n_pairs > p->n_pairs)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pair range out of bounds");
  if (camera_model != 1)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "keypoint ingest is pinhole only (KeyPoint::Unproject, keypoints.cc:59-60)");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  if (!K_inv) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "K_inv is NULL");
  if (n_pairs == 0) return 0;
  if (int rc = materialize(p)) return rc;
  const int64_t m = p->offsets[(size_t)(first_pair + n_pairs)] - p->offsets[(size_t)first_pair];
  if (m > 0 && (!pts1 || !pts2)) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pts1/pts2 is NULL");
  if (m > 0 && p->nc >= 12 && !cov2) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "cov2 is NULL for a PNEC-mode problem");
  if (m > 0 && p->nc >= 18 && !cov1) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "cov1 is NULL for a SYM-mode problem");
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  const double *d_p1 = pts1, *d_p2 = pts2, *d_c2 = cov2, *d_c1 = cov1, *d_K = K_inv;
  double *tmp = nullptr;
  if (space == PNEC_HIP_MEM_HOST) {
    const int64_t per = 4 + (p->nc >= 12 ? 3 : 0) + (p->nc >= 18 ? 3 : 0);
    PNEC_HIP_TRY(dev_alloc(&tmp, sizeof(double) * (per * m + 9)));
    double *w = tmp;
    auto up = [&](const double *src, int64_t k, const double **dst) -> hipError_t {
      *dst = w;
      hipError_t e = k ? hipMemcpyAsync(w, src, sizeof(double) * k, hipMemcpyHostToDevice, stream) : hipSuccess;
      w += k;
      return e;
    };
    hipError_t e = up(K_inv, 9, &d_K);
    if (e == hipSuccess) e = up(pts1, 2 * m, &d_p1);
    if (e == hipSuccess) e = up(pts2, 2 * m, &d_p2);
    if (e == hipSuccess && p->nc >= 12) e = up(cov2, 3 * m, &d_c2);
    if (e == hipSuccess && p->nc >= 18) e = up(cov1, 3 * m, &d_c1);
    if (e != hipSuccess) {
      (void)dev_free(tmp);
      return fail_hip(e, "hipMemcpyAsync(H2D)");
    }
  }
  const int32_t n_max = std::max<int32_t>(p->n_max, 1);
  const dim3 block(256);
  const dim3 grid((unsigned)std::min<int64_t>((n_max + 255) / 256, 64), (unsigned)std::min<int64_t>(n_pairs, 32768));
  switch (p->nc) {
    case 6:
      hipLaunchKernelGGL(ingest_keypoints_kernel<6>, grid, block, 0, stream, p->d_data, p->d_block_offset, p->d_offsets,
                         p->d_count, first_pair, n_pairs, d_p1, d_p2, d_c2, d_c1, d_K, kappa);
      break;
    case 12:
      hipLaunchKernelGGL(ingest_keypoints_kernel<12>, grid, block, 0, stream, p->d_data, p->d_block_offset, p->d_offsets,
                         p->d_count, first_pair, n_pairs, d_p1, d_p2, d_c2, d_c1, d_K, kappa);
      break;
    default:
      hipLaunchKernelGGL(ingest_keypoints_kernel<18>, grid, block, 0, stream, p->d_data, p->d_block_offset, p->d_offsets,
                         p->d_count, first_pair, n_pairs, d_p1, d_p2, d_c2, d_c1, d_K, kappa);
      break;
  }
  hipError_t e = hipGetLastError();
  if (tmp) {
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)dev_free(tmp);
  }
  if (e != hipSuccess) return fail_hip(e, "ingest_keypoints_kernel");
  return 0;
}

int64_t pnec_hip_problem_payload_doubles(const pnec_hip_problem *p) { return p ? p->data_doubles : 0; }

int pnec_hip_problem_export_payload(const pnec_hip_problem *p, double *out, int space, void *stream_) {
  if (!p || !out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  if (p->data_doubles == 0) return 0;
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  PNEC_HIP_TRY(hipMemcpyAsync(out, p->d_data, sizeof(double) * p->data_doubles,
                              space == PNEC_HIP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, stream));
  if (space == PNEC_HIP_MEM_HOST) PNEC_HIP_TRY(hipStreamSynchronize(stream));
  return 0;
}

// Host-side sizes of a batch whose real pair sizes so far exist only on the device (a batch made by
// InlierExtraction): wait for the producing stream, fetch the counts, rebuild offsets / totals.
static int materialize(const pnec_hip_problem *cp) {
  pnec_hip_problem *p = const_cast<pnec_hip_problem *>(cp);
  if (!p || !p->lazy) return 0;
  DeviceGuard guard(p->device);
  PNEC_HIP_TRY(hipStreamSynchronize(p->lazy_stream));
  if (p->n_pairs > 0)
    PNEC_HIP_TRY(hipMemcpy(p->host_counts.data(), p->d_count, sizeof(int32_t) * p->n_pairs, hipMemcpyDeviceToHost));
  p->n_max = 0;
  for (int64_t i = 0; i < p->n_pairs; ++i) {
    p->offsets[(size_t)i + 1] = p->offsets[(size_t)i] + p->host_counts[(size_t)i];
    p->n_max = std::max(p->n_max, p->host_counts[(size_t)i]);
  }
  p->n_corr = p->offsets[(size_t)p->n_pairs];
  p->lazy = false;
  // the geometry buckets were chosen from the source's sizes (upper bounds): rebuild them from the real ones
  p->buckets.clear();
  if (p->d_bucket_pairs) (void)dev_free(p->d_bucket_pairs);
  p->d_bucket_pairs = nullptr;
  return 0;
}

int64_t pnec_hip_problem_num_pairs(const pnec_hip_problem *p) { return p ? p->n_pairs : 0; }
int64_t pnec_hip_problem_num_correspondences(const pnec_hip_problem *p) {
  return p && materialize(p) == 0 ? p->n_corr : 0;
}
int64_t pnec_hip_problem_max_correspondences(const pnec_hip_problem *p) {
  return p && materialize(p) == 0 ? p->n_max : 0;
}
int64_t pnec_hip_problem_payload_bytes(const pnec_hip_problem *p) {
  return p && materialize(p) == 0 ? p->n_corr * p->nc * (int64_t)sizeof(double) : 0;
}
int pnec_hip_problem_offsets(const pnec_hip_problem *p, int64_t *out) {
  if (!p || !out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (int rc = materialize(p)) return rc;
  std::memcpy(out, p->offsets.data(), sizeof(int64_t) * p->offsets.size());
  return 0;
}
int pnec_hip_problem_mode(const pnec_hip_problem *p) { return p ? p->mode : -1; }
int pnec_hip_problem_device(const pnec_hip_problem *p) { return p ? p->device : -1; }

int pnec_hip_describe_launch(const pnec_hip_problem *p, const pnec_hip_options *opt,
                             int32_t *corr_per_lane, int32_t *waves_per_pair,
                             int32_t *lds_corr_per_lane, int32_t *threads_per_block,
                             int32_t *resident) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  Geometry g;
  if (int rc = choose_geometry(p, opt, &g)) return rc;
  if (corr_per_lane) *corr_per_lane = g.cpl;
  if (waves_per_pair) *waves_per_pair = g.wpp;
  if (lds_corr_per_lane) *lds_corr_per_lane = g.ldsk;
  if (threads_per_block) *threads_per_block = kWave * g.wpp;
  if (resident) *resident = g.resident ? 1 : 0;
  return 0;
}

int pnec_hip_solve(pnec_hip_problem *p, const double *init_q, const double *init_t, int32_t n_hyp,
                   const double *hyp_t, double reg, const pnec_hip_options *opt_in, double *out_q,
                   double *out_t, double *out_cost, int32_t *out_iterations, int32_t *out_status,
                   int space, void *stream_) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  if (!init_q) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "init_q is NULL");
  if (!hyp_t) n_hyp = 1;
  if (n_hyp < 1) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "n_hyp must be >= 1");
  if (!hyp_t && !init_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "init_t and hyp_t are both NULL");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  pnec_hip_options opt;
  if (opt_in)
    opt = *opt_in;
  else
    pnec_hip_default_options(&opt);
  if (opt.max_num_iterations < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "max_num_iterations < 0");
  if (opt.flags & ~(PNEC_HIP_OPT_COUNT_PASSES | PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "pnec_hip_options.flags: undefined bit set");
  const bool numeric = (opt.flags & PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL) != 0;
  if (numeric) {   // verification mode: the streaming form, whatever the tuning fields say
    opt.corr_per_lane = 0;
    opt.waves_per_pair = kStreamWaves;
    opt.lds_corr_per_lane = 0;
  }
  const int64_t S = p->n_pairs * (int64_t)n_hyp;
  if (S == 0) return 0;
  if (S > 0x7fffffffLL) return fail(PNEC_HIP_ERR_UNSUPPORTED, "more than 2^31-1 solves in one call");
  Geometry g;
  if (int rc = choose_geometry(p, &opt, &g)) return rc;

  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  SolveArgs a;
  std::memset(&a, 0, sizeof(a));
  a.data = p->d_data;
  a.block_offset = p->d_block_offset;
  a.count = p->d_count;
  a.n_solves = S;
  a.n_hyp = n_hyp;
  a.reg = reg;
  a.opt = opt;
  finish_args(a);
  a.numeric_jacobian = numeric ? 1 : 0;
  if (opt.flags & PNEC_HIP_OPT_COUNT_PASSES) {   // diagnostics: count the passes this call executes (pnec_hip_work_counters)
    if (int rc = solve_work_buffer(p->device, &a.work)) return rc;
  }

  if (space == PNEC_HIP_MEM_DEVICE) {
    a.init_q = init_q;
    a.init_t = init_t;
    a.hyp_t = hyp_t;
    a.out_q = out_q;
    a.out_t = out_t;
    a.out_cost = out_cost;
    a.out_iterations = out_iterations;
    a.out_status = out_status;
  } else {
    // stage: [init_q 4P | init_t 3P | hyp_t 3S | out_q 4S | out_t 3S | out_cost S], ints [it S | st S]
    const int64_t P = p->n_pairs;
    if (int rc = ensure_stage(p, 7 * P + 11 * S, 2 * S)) return rc;
    double *w = p->d_stage;
    double *s_q = w;      w += 4 * P;
    double *s_t = w;      w += 3 * P;
    double *s_h = w;      w += 3 * S;
    double *s_oq = w;     w += 4 * S;
    double *s_ot = w;     w += 3 * S;
    double *s_oc = w;
    PNEC_HIP_TRY(hipMemcpyAsync(s_q, init_q, sizeof(double) * 4 * P, hipMemcpyHostToDevice, stream));
    if (init_t)
      PNEC_HIP_TRY(hipMemcpyAsync(s_t, init_t, sizeof(double) * 3 * P, hipMemcpyHostToDevice, stream));
    if (hyp_t)
      PNEC_HIP_TRY(hipMemcpyAsync(s_h, hyp_t, sizeof(double) * 3 * S, hipMemcpyHostToDevice, stream));
    a.init_q = s_q;
    a.init_t = init_t ? s_t : nullptr;
    a.hyp_t = hyp_t ? s_h : nullptr;
    a.out_q = s_oq;
    a.out_t = s_ot;
    a.out_cost = s_oc;
    a.out_iterations = p->d_stage_i;
    a.out_status = p->d_stage_i + S;
  }

  // several hypotheses per pair on a several-wavefront geometry: the block-per-(pair, group of hypotheses) form -- the
  // pair's payload loaded once per group, one LM step for the whole group (bit-identical results).  PNEC_SOLVE_GROUPS=0
  // keeps the one-solve-per-block launch (A/B, and the bit-identity test).
  static const bool use_groups = [] {
    const char *ev = std::getenv("PNEC_SOLVE_GROUPS");
    return !(ev && *ev == '0');
  }();
  auto launch_on = [&](const Geometry &gg, const SolveArgs &aa, hipStream_t st) -> hipError_t {
    if (use_groups && aa.n_hyp > 1 && gg.resident && !aa.trace && !aa.numeric_jacobian &&
        (gg.wpp >= 2 ? group_geometry_ok(p->mode, gg.cpl, gg.wpp, gg.ldsk)
                     : pairhyp_geometry_ok(p->mode, gg.cpl, gg.wpp, gg.ldsk))) {
      switch (p->mode) {
        case PNEC_HIP_MODE_NEC: return launch_solve_group_mode_0(gg.cpl, gg.wpp, gg.ldsk, aa, st);
        case PNEC_HIP_MODE_TARGET: return launch_solve_group_mode_1(gg.cpl, gg.wpp, gg.ldsk, aa, st);
        case PNEC_HIP_MODE_HOST: return launch_solve_group_mode_2(gg.cpl, gg.wpp, gg.ldsk, aa, st);
        default: return launch_solve_group_mode_3(gg.cpl, gg.wpp, gg.ldsk, aa, st);
      }
    }
    switch (p->mode) {
      case PNEC_HIP_MODE_NEC: return launch_solve_mode_0(gg.cpl, gg.wpp, gg.ldsk, gg.resident, aa, st);
      case PNEC_HIP_MODE_TARGET: return launch_solve_mode_1(gg.cpl, gg.wpp, gg.ldsk, gg.resident, aa, st);
      case PNEC_HIP_MODE_HOST: return launch_solve_mode_2(gg.cpl, gg.wpp, gg.ldsk, gg.resident, aa, st);
      default: return launch_solve_mode_3(gg.cpl, gg.wpp, gg.ldsk, gg.resident, aa, st);
    }
  };
  auto launch = [&](const Geometry &gg, const SolveArgs &aa) -> hipError_t { return launch_on(gg, aa, stream); };
  // PNEC_HIP_TRACE=<file>: per-workgroup phase timestamps of every launch (diagnostics; adds a
  // device synchronisation, so never set it for timed runs)
  const char *trace_path = std::getenv("PNEC_HIP_TRACE");
  unsigned long long *d_trace = nullptr;
  const bool forced = opt.corr_per_lane > 0 || opt.waves_per_pair > 0;
  if (!forced) {
    if (int rc = ensure_buckets(p)) return rc;
  }
  if (trace_path && *trace_path) {
    PNEC_HIP_TRY(dev_alloc(&d_trace, sizeof(unsigned long long) * 4 * S));
    const hipError_t te = hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 4 * S, stream);
    if (te != hipSuccess) {
      (void)dev_free(d_trace);
      return fail_hip(te, "hipMemsetAsync(trace)");
    }
    a.trace = d_trace;
  }
  hipError_t e = hipSuccess;
  if (forced || p->buckets.size() <= 1) {
    e = launch(g, a);
  } else {
    // ragged batch: one launch per geometry in use, over the pairs that fit it -- side by side: the first on
    // the caller's stream, the others on streams of the batch's own that fork from it and join it again
    const size_t n_side = p->buckets.size() - 1;
    if (int rc = ensure_side_streams(p, n_side)) {
      if (d_trace) (void)dev_free(d_trace);
      return rc;
    }
    e = hipEventRecord(p->fork_event, stream);
    for (size_t b = 0; b < p->buckets.size() && e == hipSuccess; ++b) {
      const auto &bk = p->buckets[b];
      SolveArgs ab = a;
      ab.pair_index = p->d_bucket_pairs + bk.first;
      ab.n_solves = bk.count * (int64_t)n_hyp;
      if (ab.trace) ab.trace += 4 * (size_t)(bk.first * (int64_t)n_hyp);  // records are indexed by blockIdx per launch
      const Geometry gb = {bk.cpl, bk.wpp, bk.ldsk, bk.resident};
      if (b == 0) {
        e = launch_on(gb, ab, stream);
      } else {
        hipStream_t st = p->side_streams[b - 1];
        e = hipStreamWaitEvent(st, p->fork_event, 0);
        if (e == hipSuccess) e = launch_on(gb, ab, st);
        if (e == hipSuccess) e = hipEventRecord(p->side_done[b - 1], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, p->side_done[b - 1], 0);
      }
    }
  }
  if (d_trace) {
    std::vector<unsigned long long> h(4 * (size_t)S);
    hipError_t te = hipStreamSynchronize(stream);
    if (te == hipSuccess) te = hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    (void)dev_free(d_trace);
    if (te == hipSuccess) {
      if (FILE *f = std::fopen(trace_path, "ab")) {
        std::fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
        std::fclose(f);
      }
    }
  }
  if (e != hipSuccess) return fail_hip(e, "lm_solve_kernel launch");

  if (space == PNEC_HIP_MEM_HOST) {
    if (out_q) PNEC_HIP_TRY(hipMemcpyAsync(out_q, a.out_q, sizeof(double) * 4 * S, hipMemcpyDeviceToHost, stream));
    if (out_t) PNEC_HIP_TRY(hipMemcpyAsync(out_t, a.out_t, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, stream));
    if (out_cost) PNEC_HIP_TRY(hipMemcpyAsync(out_cost, a.out_cost, sizeof(double) * S, hipMemcpyDeviceToHost, stream));
    if (out_iterations)
      PNEC_HIP_TRY(hipMemcpyAsync(out_iterations, a.out_iterations, sizeof(int32_t) * S, hipMemcpyDeviceToHost, stream));
    if (out_status)
      PNEC_HIP_TRY(hipMemcpyAsync(out_status, a.out_status, sizeof(int32_t) * S, hipMemcpyDeviceToHost, stream));
    PNEC_HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

int pnec_hip_select_best(int64_t n_pairs, int32_t n_hyp, const double *cost, int32_t *best_index,
                         int space, int device, void *stream_) {
  if (n_pairs < 0 || n_hyp < 1 || !cost || !best_index)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad arguments");
  if (n_pairs == 0) return 0;
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed");
  hipStream_t stream = (hipStream_t)stream_;
  const double *d_cost = cost;
  int32_t *d_best = best_index;
  double *tmp_c = nullptr;
  int32_t *tmp_b = nullptr;
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(dev_alloc(&tmp_c, sizeof(double) * n_pairs * n_hyp));
    hipError_t e = dev_alloc(&tmp_b, sizeof(int32_t) * n_pairs);
    if (e == hipSuccess)
      e = hipMemcpyAsync(tmp_c, cost, sizeof(double) * n_pairs * n_hyp, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      (void)dev_free(tmp_c);
      if (tmp_b) (void)dev_free(tmp_b);
      return fail_hip(e, "select_best staging");
    }
    d_cost = tmp_c;
    d_best = tmp_b;
  }
  hipLaunchKernelGGL(select_best_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0,
                     stream, n_pairs, (int)n_hyp, d_cost, d_best);
  hipError_t e = hipGetLastError();
  if (space == PNEC_HIP_MEM_HOST) {
    if (e == hipSuccess)
      e = hipMemcpyAsync(best_index, tmp_b, sizeof(int32_t) * n_pairs, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)dev_free(tmp_c);
    (void)dev_free(tmp_b);
  }
  if (e != hipSuccess) return fail_hip(e, "select_best_kernel");
  return 0;
}

int pnec_hip_cost_function(pnec_hip_problem *p, const double *q, const double *t, double *out,
                           int space, void *stream_) {
  if (!p || !q || !t || !out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (p->mode != PNEC_HIP_MODE_TARGET)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "cost_function needs a TARGET-mode problem");
  if (p->n_pairs == 0) return 0;
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  const double *d_q = q, *d_t = t;
  double *d_out = out;
  const int64_t P = p->n_pairs;
  if (space == PNEC_HIP_MEM_HOST) {
    if (int rc = ensure_stage(p, 8 * P, 0)) return rc;
    PNEC_HIP_TRY(hipMemcpyAsync(p->d_stage, q, sizeof(double) * 4 * P, hipMemcpyHostToDevice, stream));
    PNEC_HIP_TRY(hipMemcpyAsync(p->d_stage + 4 * P, t, sizeof(double) * 3 * P, hipMemcpyHostToDevice, stream));
    d_q = p->d_stage;
    d_t = p->d_stage + 4 * P;
    d_out = p->d_stage + 7 * P;
  }
  hipLaunchKernelGGL(cost_function_kernel, dim3((unsigned)P), dim3(kWave), 0, stream, p->d_data,
                     p->d_block_offset, p->d_count, d_q, d_t, d_out);
  PNEC_HIP_TRY(hipGetLastError());
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(hipMemcpyAsync(out, d_out, sizeof(double) * P, hipMemcpyDeviceToHost, stream));
    PNEC_HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

// shared driver of the two eigensolver stages (host or device pointers)
static int run_front_stage(pnec_hip_problem *p, bool weighted, const double *init_q, const double *init_t,
                           double reg, int weighted_iterations, double *out_q, double *out_t, int space,
                           void *stream_) {
  if (!p || !init_q || !out_q || !out_t || (weighted && !init_t))
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (weighted && p->mode != PNEC_HIP_MODE_TARGET)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "the weighted eigensolver needs a TARGET-mode problem");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  if (p->n_pairs == 0) return 0;
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t P = p->n_pairs;
  const double *d_q = init_q, *d_t = init_t;
  double *d_oq = out_q, *d_ot = out_t;
  if (space == PNEC_HIP_MEM_HOST) {
    if (int rc = ensure_stage(p, 14 * P, 0)) return rc;
    double *w = p->d_stage;
    PNEC_HIP_TRY(hipMemcpyAsync(w, init_q, sizeof(double) * 4 * P, hipMemcpyHostToDevice, stream));
    d_q = w; w += 4 * P;
    if (init_t) PNEC_HIP_TRY(hipMemcpyAsync(w, init_t, sizeof(double) * 3 * P, hipMemcpyHostToDevice, stream));
    d_t = w; w += 3 * P;
    d_oq = w; w += 4 * P;
    d_ot = w;
  }
  if (int rc = ensure_front(p)) return rc;
  hipError_t e = weighted
                     ? launch_weighted_eigensolver(p->device, p->d_data, p->d_block_offset, p->d_count, P, p->n_max, d_q,
                                                   d_t, reg, weighted_iterations, d_oq, d_ot, nullptr, p->d_front,
                                                   p->d_front_i, stream, p->es_scheme)
                     : launch_nec_eigensolver(p->d_data, p->d_block_offset, p->d_count, P, d_q, d_oq, d_ot,
                                              nullptr, p->d_front, p->d_front_i, stream, p->es_scheme);
  if (e != hipSuccess) return fail_hip(e, weighted ? "weighted_eigensolver_kernel" : "nec_eigensolver_kernel");
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(hipMemcpyAsync(out_q, d_oq, sizeof(double) * 4 * P, hipMemcpyDeviceToHost, stream));
    PNEC_HIP_TRY(hipMemcpyAsync(out_t, d_ot, sizeof(double) * 3 * P, hipMemcpyDeviceToHost, stream));
    PNEC_HIP_TRY(hipStreamSynchronize(stream));
  }
  return 0;
}

int pnec_hip_problem_set_eigensolver_scheme(pnec_hip_problem *p, int32_t scheme) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  if (scheme < PNEC_HIP_ES_NEWTON || scheme > PNEC_HIP_ES_LM)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "eigensolver scheme: 0 (Newton), 1 (descent) or 2 (LM)");
  p->es_scheme = scheme;
  if (p->sel_view) p->sel_view->es_scheme = scheme;   // a view handed out earlier follows its source
  return 0;
}
int pnec_hip_problem_eigensolver_scheme(const pnec_hip_problem *p) { return p ? p->es_scheme : 0; }

int pnec_hip_problem_set_ransac_flags(pnec_hip_problem *p, int32_t flags) {
  if (!p) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "problem is NULL");
  if (flags & ~PNEC_HIP_RANSAC_CHAINED_STARTS) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "unknown RANSAC flag");
  p->ransac_flags = flags;
  return 0;
}
int pnec_hip_problem_ransac_flags(const pnec_hip_problem *p) { return p ? p->ransac_flags : 0; }

int pnec_hip_nec_eigensolver(pnec_hip_problem *p, const double *init_q, double *out_q, double *out_t,
                             int space, void *stream) {
  return run_front_stage(p, false, init_q, nullptr, 0.0, 0, out_q, out_t, space, stream);
}

int pnec_hip_weighted_eigensolver(pnec_hip_problem *p, const double *init_q, const double *init_t,
                                  double reg, int32_t weighted_iterations, double *out_q, double *out_t,
                                  int space, void *stream) {
  if (weighted_iterations < 0) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "weighted_iterations < 0");
  // DESCENT moves the rotation in every round and keeps a minimiser per round (kEsMaxRounds of them); NEWTON and LM
  // chain calls only while one ends at its evaluation cap, and freeze a pair's rotation after kEsMaxRounds such calls
  if (p && p->es_scheme == PNEC_HIP_ES_DESCENT && weighted_iterations - 1 > kEsMaxRounds)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "eigensolver scheme 1 (descent) holds at most 16 weighted_iterations");
  return run_front_stage(p, true, init_q, init_t, reg, weighted_iterations, out_q, out_t, space, stream);
}

int pnec_hip_ransac_eigensolver(pnec_hip_problem *p, const double *init_q, uint64_t seed,
                                int32_t max_iterations, int32_t sample_size, double threshold, double *out_q,
                                double *out_t, uint8_t *out_inlier_mask, int32_t *out_inlier_count,
                                int32_t *out_ransac_iterations, int space, void *stream_) {
  if (!p || !init_q || !out_q || !out_t) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (max_iterations < 0 || sample_size < 1) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad RANSAC parameters");
  if (sample_size > PNEC_HIP_MAX_RANSAC_SAMPLE)
    return fail(PNEC_HIP_ERR_UNSUPPORTED, "ransac sample_size > 16 is not built (a hypothesis keeps its sample in registers)");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  if (p->n_pairs == 0) return 0;
  if (space == PNEC_HIP_MEM_HOST)  // host-side per-correspondence arrays need the exact sizes
    if (int rc = materialize(p)) return rc;
  DeviceGuard guard(p->device);
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t P = p->n_pairs, M = p->n_corr;
  const double *d_q = init_q;
  double *d_oq = out_q, *d_ot = out_t;
  uint8_t *d_mask = out_inlier_mask;
  int32_t *d_cnt = out_inlier_count, *d_it = out_ransac_iterations;
  uint8_t *tmp_mask = nullptr;
  if (space == PNEC_HIP_MEM_HOST) {
    if (int rc = ensure_stage(p, 11 * P, 2 * P)) return rc;
    double *w = p->d_stage;
    PNEC_HIP_TRY(hipMemcpyAsync(w, init_q, sizeof(double) * 4 * P, hipMemcpyHostToDevice, stream));
    d_q = w; w += 4 * P;
    d_oq = w; w += 4 * P;
    d_ot = w;
    d_cnt = p->d_stage_i;
    d_it = p->d_stage_i + P;
    if (out_inlier_mask) {
      PNEC_HIP_TRY(dev_alloc(&tmp_mask, (size_t)std::max<int64_t>(M, 1)));
      d_mask = tmp_mask;
    }
  }
  int rc_ws = ensure_front(p);
  if (!rc_ws) rc_ws = ensure_order_hint(p);
  if (rc_ws) {
    if (tmp_mask) (void)dev_free(tmp_mask);
    return rc_ws;
  }
  if (p->order_hint && !d_it) d_it = p->d_hint_its;
  hipError_t e = launch_ransac_eigensolver(p->d_data, p->d_block_offset, p->d_offsets, p->d_count, P, d_q, seed,
                                           /*first_pair_id*/ 0ull, max_iterations, sample_size, threshold, d_oq, d_ot, d_mask, d_cnt, d_it,
                                           p->d_front, p->d_front_i, stream, nullptr, nullptr, nullptr, 0, nullptr, nullptr,
                                           nullptr, nullptr,
                                           p->order_hint && p->order_pairs == P ? p->d_order : nullptr, p->es_scheme,
                                           p->ransac_flags);
  if (e == hipSuccess && p->order_hint) {   // the next call's launch order from this call's counts
    e = launch_ransac_order(d_it, P, p->d_order, stream);
    p->order_pairs = P;
  }
  if (e == hipSuccess && space == PNEC_HIP_MEM_HOST) {
    e = hipMemcpyAsync(out_q, d_oq, sizeof(double) * 4 * P, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_t, d_ot, sizeof(double) * 3 * P, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && out_inlier_mask && M > 0)
      e = hipMemcpyAsync(out_inlier_mask, d_mask, (size_t)M, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && out_inlier_count)
      e = hipMemcpyAsync(out_inlier_count, d_cnt, sizeof(int32_t) * P, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && out_ransac_iterations)
      e = hipMemcpyAsync(out_ransac_iterations, d_it, sizeof(int32_t) * P, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  if (tmp_mask) (void)dev_free(tmp_mask);
  if (e != hipSuccess) return fail_hip(e, "ransac_eigensolver_kernel");
  return 0;
}

// A batch with the capacity (block layout) of `src` and no contents yet: the target of InlierExtraction.
static int alloc_like(pnec_hip_problem *src, hipStream_t stream, pnec_hip_problem **out) {
  *out = nullptr;
  pnec_hip_problem *d = new (std::nothrow) pnec_hip_problem();
  if (!d) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "out of host memory");
  d->device = src->device;
  d->mode = src->mode;
  d->nc = src->nc;
  d->es_scheme = src->es_scheme;
  d->n_pairs = src->n_pairs;
  d->n_corr = src->n_corr;          // upper bounds until materialize()
  d->n_max = src->n_max;
  d->host_counts = src->host_counts;
  d->offsets = src->offsets;
  d->data_doubles = src->data_doubles;
  // as roomy as the source can ever get, so that a cached view survives the source's re-shaping
  d->cap_pairs = std::max(src->cap_pairs, src->n_pairs);
  d->cap_doubles = std::max(src->cap_doubles, src->data_doubles);
  const int64_t P = std::max<int64_t>(d->cap_pairs, 1);
  hipError_t e = dev_alloc(&d->d_data, sizeof(double) * (std::max<int64_t>(d->cap_doubles, 1) + kDataSlackDoubles));
  if (e == hipSuccess) e = dev_alloc(&d->d_block_offset, sizeof(int64_t) * P);
  if (e == hipSuccess) e = dev_alloc(&d->d_offsets, sizeof(int64_t) * (P + 1));
  if (e == hipSuccess) e = dev_alloc(&d->d_count, sizeof(int32_t) * P);
  if (e == hipSuccess && src->n_pairs > 0)
    e = hipMemcpyAsync(d->d_block_offset, src->d_block_offset, sizeof(int64_t) * src->n_pairs,
                       hipMemcpyDeviceToDevice, stream);
  d->view_src_gen = src->layout_gen;
  if (e != hipSuccess) {
    pnec_hip_problem_destroy(d);
    return fail_hip(e, "InlierExtraction target allocation");
  }
  *out = d;
  return 0;
}

// PNEC::InlierExtraction on the device, nothing read back: counts by ballot, offsets by a scan, the kept
// correspondences compacted pair by pair into dst (which has src's capacity).  All on `stream`.
// known_counts (optional, device): the inliers per pair when the producer of the mask counted them already (RANSAC
// does): the counting launch is skipped
// select_prepare: dst follows the source's current shape (block layout by generation); select_finish: the offsets of the
// kept correspondences + the host-side bookkeeping.  Between the two something fills dst's planes and counts: the copy
// kernel below (select_into), or the RANSAC stage itself (the chain: InlierExtraction fused into a pair's last pass).
static int select_prepare(pnec_hip_problem *src, hipStream_t stream, pnec_hip_problem *dst) {
  const int64_t P = src->n_pairs;
  if (dst->view_src_gen != src->layout_gen) {  // the source has been re-shaped since dst copied its block layout
    if (P > 0)
      PNEC_HIP_TRY(hipMemcpyAsync(dst->d_block_offset, src->d_block_offset, sizeof(int64_t) * P, hipMemcpyDeviceToDevice,
                                  stream));
    dst->view_src_gen = src->layout_gen;
    dst->n_pairs = P;
    dst->data_doubles = src->data_doubles;
    dst->buckets.clear();
    if (dst->d_bucket_pairs) (void)dev_free(dst->d_bucket_pairs);
    dst->d_bucket_pairs = nullptr;
    dst->lazy = false;  // (so that select_finish re-installs the source's bounds)
    dst->offsets = src->offsets;
  }
  return 0;
}
static int select_finish(pnec_hip_problem *src, hipStream_t stream, pnec_hip_problem *dst, bool scan = true) {
  const int64_t P = src->n_pairs;
  if (scan && P > 1) {
    hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, stream, dst->d_count, dst->d_offsets, P);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "offsets_scan_kernel");
  }
  if (!dst->lazy) {  // it had been given exact sizes: back to the source's bounds, buckets included
    dst->buckets.clear();
    if (dst->d_bucket_pairs) (void)dev_free(dst->d_bucket_pairs);
    dst->d_bucket_pairs = nullptr;
    dst->offsets = src->offsets;
  }
  dst->lazy = true;
  dst->lazy_stream = stream;
  dst->n_corr = src->n_corr;
  dst->n_max = src->n_max;
  dst->n_pairs = P;
  dst->data_doubles = src->data_doubles;  // (also when the block layout is unchanged but the pair's size is not)
  if (dst->host_counts != src->host_counts) {  // a re-shaped source: the launch geometries follow its new bounds
    dst->host_counts = src->host_counts;
    dst->offsets = src->offsets;
    dst->buckets.clear();
    if (dst->d_bucket_pairs) (void)dev_free(dst->d_bucket_pairs);
    dst->d_bucket_pairs = nullptr;
  }
  return 0;
}

static int select_into(pnec_hip_problem *src, const uint8_t *d_mask, hipStream_t stream, pnec_hip_problem *dst,
                       const int32_t *known_counts = nullptr) {
  const int64_t P = src->n_pairs;
  if (int rc = select_prepare(src, stream, dst)) return rc;
  if (P > 0) {
    hipError_t e = hipSuccess;
    if (known_counts) {
      // (the copy kernel also installs the counts in dst and, for a batch of one pair, its AoS offsets; the scan
      // of a larger batch's counts follows it: nothing in the copy needs the new offsets)
      e = launch_select(src->nc, src->d_data, src->d_block_offset, src->d_offsets, src->d_count, d_mask, dst->d_data,
                        dst->d_block_offset, known_counts, dst->d_count, P == 1 ? dst->d_offsets : (int64_t *)nullptr, P,
                        stream);
      if (e == hipSuccess && P > 1) {
        hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, stream, dst->d_count, dst->d_offsets, P);
        e = hipGetLastError();
      }
    } else {
      hipLaunchKernelGGL(mask_count_kernel, dim3((unsigned)P), dim3(kWave), 0, stream, d_mask, src->d_offsets,
                         src->d_count, dst->d_count, P == 1 ? dst->d_offsets : (int64_t *)nullptr);
      if (P > 1) hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(1024), 0, stream, dst->d_count, dst->d_offsets, P);
      e = hipGetLastError();
      if (e == hipSuccess)
        e = launch_select(src->nc, src->d_data, src->d_block_offset, src->d_offsets, src->d_count, d_mask, dst->d_data,
                          dst->d_block_offset, dst->d_count, dst->d_count, nullptr, P, stream);
    }
    if (e != hipSuccess) return fail_hip(e, "select_kernel");
  }
  return select_finish(src, stream, dst, /*scan*/ false);  // (the scans are among the launches above)
}

int pnec_hip_problem_select(pnec_hip_problem *src, const uint8_t *mask, int space, void *stream_,
                            pnec_hip_problem **out) {
  if (!src || !mask || !out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  *out = nullptr;
  DeviceGuard guard(src->device);
  hipStream_t stream = (hipStream_t)stream_;
  // a host mask is in the caller's correspondence order: its length is the source's exact total
  if (space == PNEC_HIP_MEM_HOST)
    if (int rc = materialize(src)) return rc;
  pnec_hip_problem *dst = nullptr;
  if (int rc = alloc_like(src, stream, &dst)) return rc;
  const uint8_t *d_mask = mask;
  if (space == PNEC_HIP_MEM_HOST) {
    dst->mask_bytes = std::max<int64_t>(src->n_corr, 1);
    hipError_t e = dev_alloc(&dst->d_mask, (size_t)dst->mask_bytes);
    if (e == hipSuccess && src->n_corr > 0)
      e = hipMemcpyAsync(dst->d_mask, mask, (size_t)src->n_corr, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      pnec_hip_problem_destroy(dst);
      return fail_hip(e, "mask upload");
    }
    d_mask = dst->d_mask;
  }
  if (int rc = select_into(src, d_mask, stream, dst)) {
    pnec_hip_problem_destroy(dst);
    return rc;
  }
  if (space == PNEC_HIP_MEM_HOST) {  // HOST space calls block (the caller may reuse `mask` right away)
    const hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
      pnec_hip_problem_destroy(dst);
      return fail_hip(e, "select_kernel");
    }
  }
  *out = dst;
  return 0;
}

// the refinement's pass counters (PNEC_HIP_OPT_COUNT_PASSES): two 64-bit sums per device, allocated on first use
static unsigned long long *g_solve_work[64] = {nullptr};
static int solve_work_buffer(int device, unsigned long long **out) {
  if (device < 0 || device >= 64) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "device index out of range");
  std::lock_guard<std::mutex> lock(g_mem_mutex);
  if (!g_solve_work[device]) {
    unsigned long long *w = nullptr;
    PNEC_HIP_TRY(hipMalloc(&w, 2 * sizeof(unsigned long long)));
    PNEC_HIP_TRY(hipMemset(w, 0, 2 * sizeof(unsigned long long)));
    g_solve_work[device] = w;
  }
  *out = g_solve_work[device];
  return 0;
}

int pnec_hip_work_counters(int device, int reset, uint64_t *out16, int32_t *compiled_in) {
  if (!out16) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out16 is NULL");
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed (no such device?)");
  unsigned long long c[16];
  int in = 0;
  PNEC_HIP_TRY(frontend_work_counters(reset, c, &in));
  for (int i = 0; i < 16; ++i) out16[i] = (uint64_t)c[i];
  if (compiled_in) *compiled_in = in;
  // [13], [14]: correspondence-passes the refinement executed in full / cost-only, for calls made with
  // PNEC_HIP_OPT_COUNT_PASSES set (any build)
  if (device >= 0 && device < 64 && g_solve_work[device]) {
    unsigned long long w[2] = {0, 0};
    PNEC_HIP_TRY(hipDeviceSynchronize());
    PNEC_HIP_TRY(hipMemcpy(w, g_solve_work[device], sizeof(w), hipMemcpyDeviceToHost));
    out16[13] = w[0];
    out16[14] = w[1];
    if (reset) PNEC_HIP_TRY(hipMemset(g_solve_work[device], 0, sizeof(w)));
  }
  return 0;
}

int pnec_hip_problem_select_view(pnec_hip_problem *src, const uint8_t *mask, int space, void *stream_,
                                 pnec_hip_problem **out) {
  if (!src || !mask || !out) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (space != PNEC_HIP_MEM_DEVICE && space != PNEC_HIP_MEM_HOST)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  *out = nullptr;
  DeviceGuard guard(src->device);
  hipStream_t stream = (hipStream_t)stream_;
  if (space == PNEC_HIP_MEM_HOST)
    if (int rc = materialize(src)) return rc;  // a host mask is sized by the source's exact total
  if (!src->sel_view || src->sel_view->cap_doubles < src->data_doubles || src->sel_view->cap_pairs < src->n_pairs) {
    if (src->sel_view) pnec_hip_problem_destroy(src->sel_view);
    src->sel_view = nullptr;
    if (int rc = alloc_like(src, stream, &src->sel_view)) return rc;
  }
  const uint8_t *d_mask = mask;
  if (space == PNEC_HIP_MEM_HOST) {
    const int64_t M = std::max<int64_t>(src->n_corr, 1);
    if (src->mask_bytes < M) {
      if (src->d_mask) (void)dev_free(src->d_mask);
      src->d_mask = nullptr;
      src->mask_bytes = 0;
      const int64_t want = std::max<int64_t>(M, src->cap_doubles / std::max(src->nc, 1));
      PNEC_HIP_TRY(dev_alloc(&src->d_mask, (size_t)want));
      src->mask_bytes = want;
    }
    if (src->n_corr > 0)
      PNEC_HIP_TRY(hipMemcpyAsync(src->d_mask, mask, (size_t)src->n_corr, hipMemcpyHostToDevice, stream));
    d_mask = src->d_mask;
  }
  if (int rc = select_into(src, d_mask, stream, src->sel_view)) return rc;
  // the view runs the stages the way its source would NOW (the scheme may have been changed since the view was made)
  src->sel_view->es_scheme = src->es_scheme;
  if (space == PNEC_HIP_MEM_HOST) PNEC_HIP_TRY(hipStreamSynchronize(stream));  // (the caller may reuse `mask`)
  *out = src->sel_view;
  return 0;
}

int pnec_hip_unscented_transform(int64_t n, const double *mu, const double *covs, const double *K_inv,
                                 double kappa, int camera_model, double *out_bvs, double *out_covs,
                                 int space, int device, void *stream_) {
  if (n < 0 || (n > 0 && (!mu || !covs || !out_covs)) || !K_inv)
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "NULL argument");
  if (camera_model != 0 && camera_model != 1) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "camera_model must be 0 or 1");
  if (n == 0) return 0;
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed (no such device?)");
  hipStream_t stream = (hipStream_t)stream_;
  const double *d_mu = mu, *d_cov = covs, *d_K = K_inv;
  double *d_ob = out_bvs, *d_oc = out_covs, *tmp = nullptr;
  if (space == PNEC_HIP_MEM_HOST) {
    PNEC_HIP_TRY(dev_alloc(&tmp, sizeof(double) * (24 * n + 9)));
    double *w = tmp;
    hipError_t e = hipMemcpyAsync(w, mu, sizeof(double) * 3 * n, hipMemcpyHostToDevice, stream);
    d_mu = w; w += 3 * n;
    if (e == hipSuccess) e = hipMemcpyAsync(w, covs, sizeof(double) * 9 * n, hipMemcpyHostToDevice, stream);
    d_cov = w; w += 9 * n;
    if (e == hipSuccess) e = hipMemcpyAsync(w, K_inv, sizeof(double) * 9, hipMemcpyHostToDevice, stream);
    d_K = w; w += 9;
    d_oc = w; w += 9 * n;
    d_ob = out_bvs ? w : nullptr;
    if (e != hipSuccess) {
      (void)dev_free(tmp);
      return fail_hip(e, "unscented_transform staging");
    }
  } else if (space != PNEC_HIP_MEM_DEVICE) {
    return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "bad memory space");
  }
  hipLaunchKernelGGL(unscented_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, n, d_mu,
                     d_cov, d_K, kappa, camera_model, d_ob, d_oc);
  hipError_t e = hipGetLastError();
  if (tmp) {
    if (e == hipSuccess) e = hipMemcpyAsync(out_covs, d_oc, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && out_bvs)
      e = hipMemcpyAsync(out_bvs, d_ob, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)dev_free(tmp);
  }
  if (e != hipSuccess) return fail_hip(e, "unscented_kernel");
  return 0;
}

int pnec_hip_selftest(int device) {
  DeviceGuard guard(device);
  if (!guard.ok) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "hipSetDevice failed");
  {  // the front stages' smallest-eigenpair route against the Jacobi sweeps
    double *d = nullptr;
    PNEC_HIP_TRY(dev_alloc(&d, sizeof(double) * 192));
    double h[192];
    hipError_t e = launch_frontend_selftest(d, 0);
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)dev_free(d);
    if (e != hipSuccess) return fail_hip(e, "eig_selftest_kernel");
    for (int i = 0; i < kWave; ++i) {
      char buf[160];
      if (!(h[i] < 1e-14)) {
        std::snprintf(buf, sizeof(buf), "smallest eigenpair: residual %.3g |M| in lane %d", h[i], i);
        return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
      }
      if (!(h[64 + i] < 1e-14)) {
        std::snprintf(buf, sizeof(buf), "smallest eigenpair: eigenvalue %.3g |M| above the sweeps' smallest in lane %d", h[64 + i], i);
        return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
      }
    }
  }
  double *d = nullptr;
  PNEC_HIP_TRY(dev_alloc(&d, sizeof(double) * 352));
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(kWave), 0, 0, d);
  double h[352];
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(double) * 344, hipMemcpyDeviceToHost);
  (void)dev_free(d);
  if (e != hipSuccess) return fail_hip(e, "selftest_kernel");
  for (int i = 0; i < kWave; ++i)
    if (h[i] != 89440.0) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "wave_allreduce_sum: lane %d holds %.17g, expected 89440", i, h[i]);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
  if (!(h[64] >= 0.0 && h[64] < 1e-12)) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "chol_solve5 residual too large");
  for (int i = 0; i < kWave; ++i) {
    char buf[160];
    if (!(h[216 + i] >= 0.0 && h[216 + i] < 1e-13)) {
      std::snprintf(buf, sizeof(buf), "gj_solve5_rows: lane %d deviates from chol_solve5 by %.3g (relative)", i, h[216 + i]);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
    if (h[280 + i] != 1011.0 + 16.0 * (i / 16)) {
      std::snprintf(buf, sizeof(buf), "bcast_row<11>: lane %d holds %.17g", i, h[280 + i]);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
  }
  if (!(std::abs(h[65]) < 1e-15)) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "fast_rsqrt inaccurate");
  if (!(std::abs(h[66]) < 1e-15)) return fail(PNEC_HIP_ERR_HIP_RUNTIME, "fast_rcp inaccurate");
  for (int j = 0; j < kNumAcc; ++j)
    if (h[67 + j] != 2080.0 * (j + 1) + 64.0 * j) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "wave_reduce21: sum %d is %.17g, expected %.17g", j, h[67 + j],
                    2080.0 * (j + 1) + 64.0 * j);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
  for (int i = 0; i < kWave; ++i)
    if (!(h[88 + i] < 4e-16)) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "sincos_bounded deviates from libm by %.3g", h[88 + i]);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
  for (int i = 0; i < kWave; ++i)
    if (!(h[152 + i] < 9e-16)) {
      char buf[128];
      std::snprintf(buf, sizeof(buf), "acos_lean / atan2_lean deviate from libm by %.3g", h[152 + i]);
      return fail(PNEC_HIP_ERR_HIP_RUNTIME, buf);
    }
  return 0;
}

#include "pnec_pipeline.inl"
#include "pnec_stream.inl"
#include "pnec_frame.inl"
#include "pnec_multi.inl"

int pnec_hip_alloc_counters(uint64_t *out4) {
  if (!out4) return fail(PNEC_HIP_ERR_INVALID_ARGUMENT, "out is NULL");
  std::lock_guard<std::mutex> lock(g_mem_mutex);
  out4[0] = g_n_hip_malloc;
  out4[1] = g_n_cache_hit;
  out4[2] = (uint64_t)g_live.size();
  out4[3] = (uint64_t)g_cached_bytes;
  return 0;
}

int64_t pnec_hip_release_cache(int device) {
  std::lock_guard<std::mutex> lock(g_mem_mutex);
  const size_t before = g_cached_bytes;
  release_cache_locked(device);
  release_stream_pool_locked(device);
  return (int64_t)(before - g_cached_bytes);
}

}  // extern "C"
