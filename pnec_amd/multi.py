"""Several GPUs of one node from ONE process: the persistent multi-device handle of the C ABI (include/pnec_hip.h
pnec_hip_multi_*).  Host arrays in the reference layout in, host arrays out; every listed device holds a contiguous range
of the pairs balanced by correspondence count and keeps its batch, stream and scratch across calls."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


class MultiBatch:
    """pnec_hip_multi: fill(offsets, bvs1, bvs2, covs) then solve(...) / solve_pipeline(...) as often as wanted."""

    def __init__(self, devices, mode: int, max_pairs: int, max_corr: int, max_pair_corr: int):
        self._lib = capi.lib()
        self.devices = np.ascontiguousarray(devices, dtype=np.int32)
        self.mode = int(mode)
        self._h = C.c_void_p()
        capi.check(self._lib.pnec_hip_multi_create(len(self.devices), self.devices.ctypes.data, self.mode, int(max_pairs),
                                                   int(max_corr), int(max_pair_corr), C.byref(self._h)))
        self.n_pairs = 0
        self.offsets = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pnec_hip_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def bounds(self) -> np.ndarray:
        b = np.zeros(len(self.devices) + 1, dtype=np.int64)
        capi.check(self._lib.pnec_hip_multi_bounds(self._h, b.ctypes.data))
        return b

    def fill(self, offsets, bvs1, bvs2, covs=None, covs_host=None):
        off = _c(offsets, np.int64)
        self.n_pairs, self.offsets = len(off) - 1, off
        b1, b2 = _c(bvs1).reshape(-1, 3), _c(bvs2).reshape(-1, 3)
        cv = None if covs is None else _c(np.asarray(covs).reshape(-1, 3, 3).transpose(0, 2, 1)).reshape(-1, 9)
        ch = None if covs_host is None else _c(np.asarray(covs_host).reshape(-1, 3, 3).transpose(0, 2, 1)).reshape(-1, 9)
        capi.check(self._lib.pnec_hip_multi_fill(self._h, self.n_pairs, off.ctypes.data, b1.ctypes.data, b2.ctypes.data,
                                                 None if cv is None else cv.ctypes.data, None if ch is None else ch.ctypes.data))

    def solve(self, init_q, init_t=None, reg: float = 1e-13, options: capi.Options | None = None, hyp_t=None, n_hyp: int = 1):
        """PNECCeres::Optimize for every (pair, hypothesis) -> dict(q, t, cost, iterations, status)"""
        P, H = self.n_pairs, (int(n_hyp) if hyp_t is not None else 1)
        q0 = _c(init_q).reshape(P, 4)
        t0 = None if init_t is None else _c(init_t).reshape(P, 3)
        ht = None if hyp_t is None else _c(hyp_t).reshape(P * H, 3)
        o = options if options is not None else capi.default_options()
        q, t, cost = np.zeros((P * H, 4)), np.zeros((P * H, 3)), np.zeros(P * H)
        its, st = np.zeros(P * H, dtype=np.int32), np.zeros(P * H, dtype=np.int32)
        capi.check(self._lib.pnec_hip_multi_solve(self._h, q0.ctypes.data, None if t0 is None else t0.ctypes.data, H,
                                                  None if ht is None else ht.ctypes.data, reg, C.byref(o), q.ctypes.data,
                                                  t.ctypes.data, cost.ctypes.data, its.ctypes.data, st.ctypes.data))
        return dict(q=q, t=t, cost=cost, iterations=its, status=st)

    def solve_pipeline(self, init_q, init_t, options: capi.PipelineOptions | None = None, want_inliers: bool = False):
        """PNEC::Solve for every pair -> (q, t) or (q, t, inlier_mask, inlier_count)"""
        P = self.n_pairs
        q0, t0 = _c(init_q).reshape(P, 4), _c(init_t).reshape(P, 3)
        o = options if options is not None else capi.default_pipeline_options()
        q, t = np.zeros((P, 4)), np.zeros((P, 3))
        mask = np.zeros(max(int(self.offsets[-1]), 1), dtype=np.uint8) if want_inliers else None
        cnt = np.zeros(max(P, 1), dtype=np.int32) if want_inliers else None
        capi.check(self._lib.pnec_hip_multi_solve_pipeline(self._h, q0.ctypes.data, t0.ctypes.data, C.byref(o), q.ctypes.data,
                                                           t.ctypes.data, None if mask is None else mask.ctypes.data,
                                                           None if cnt is None else cnt.ctypes.data))
        return (q, t, mask[: int(self.offsets[-1])], cnt[:P]) if want_inliers else (q, t)


def alloc_counters() -> dict:
    out = np.zeros(4, dtype=np.uint64)
    capi.check(capi.lib().pnec_hip_alloc_counters(out.ctypes.data))
    return dict(hip_malloc_calls=int(out[0]), cache_hits=int(out[1]), live_blocks=int(out[2]), cached_bytes=int(out[3]))
