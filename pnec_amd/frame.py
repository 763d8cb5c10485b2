"""Per-frame front-end over ``pnec_hip_frame_*``: PNEC::Solve -- the WHOLE chain -- for one frame pair per call,
the way the reference's odometry calls it (Frame2Frame::PNECAlign, frame2frame.cc:122-141 -> pnec.cc:77-124),
on a handle that owns everything a frame needs (pinned staging, a capacity-shaped one-pair batch, a HIP stream):
nothing is allocated per call.  numpy in, numpy out.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class FrameSolver:
    def __init__(self, max_corr: int = 4096, device: int = 0):
        self._lib = capi.lib()
        self.device = int(device)
        h = C.c_void_p()
        capi.check(self._lib.pnec_hip_frame_create(self.device, int(max_corr), None, C.byref(h)))
        self._h = h

    @property
    def capacity(self) -> int:
        return int(self._lib.pnec_hip_frame_capacity(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pnec_hip_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def solve_raw(self, n: int, bvs1, bvs2, covs9, init_q, init_t, options, out_q, out_t, out_mask) -> int:
        """The same call without any marshalling, for per-frame loops: every array is float64 (uint8 for the mask),
        C-contiguous and of the right shape already -- bvs* [n,3], covs9 [n,9] COLUMN-MAJOR 3x3 (or None), init_q [4],
        init_t [3], out_q [4], out_t [3], out_mask [>= n]; `options` a capi.PipelineOptions.  Returns the inlier count."""
        cnt = C.c_int32(0)
        capi.check(self._lib.pnec_hip_frame_solve(self._h, n, bvs1.ctypes.data, bvs2.ctypes.data,
                                                  None if covs9 is None else covs9.ctypes.data, init_q.ctypes.data,
                                                  init_t.ctypes.data, C.byref(options), out_q.ctypes.data, out_t.ctypes.data,
                                                  out_mask.ctypes.data, C.byref(cnt)))
        return int(cnt.value)

    def solve(self, bvs1, bvs2, covs, init_q, init_t, options: capi.PipelineOptions | None = None):
        """-> (q [4] xyzw, t [3], inlier_mask [n] bool, inlier_count).  bvs* [n,3]; covs [n,3,3] (symmetric) or
        [n,9] column-major, or None with use_nec; init_q xyzw."""
        b1 = np.ascontiguousarray(bvs1, dtype=np.float64).reshape(-1, 3)
        b2 = np.ascontiguousarray(bvs2, dtype=np.float64).reshape(-1, 3)
        n = len(b1)
        if len(b2) != n:
            raise ValueError("bvs1 and bvs2 differ in length")
        cv = None
        if covs is not None:
            cv = np.asarray(covs, dtype=np.float64)
            if cv.ndim == 3:
                cv = np.transpose(cv, (0, 2, 1))
            cv = np.ascontiguousarray(cv.reshape(-1, 9))
            if len(cv) != n:
                raise ValueError("covs and bvs differ in length")
        q0 = np.ascontiguousarray(init_q, dtype=np.float64).reshape(4)
        t0 = np.ascontiguousarray(init_t, dtype=np.float64).reshape(3)
        q, t = np.empty(4), np.empty(3)
        mask = np.zeros(max(n, 1), dtype=np.uint8)
        cnt = C.c_int32(0)
        p = lambda a: None if a is None or a.size == 0 else a.ctypes.data
        capi.check(self._lib.pnec_hip_frame_solve(self._h, n, p(b1), p(b2), p(cv), q0.ctypes.data, t0.ctypes.data,
                                                  C.byref(options) if options is not None else None, q.ctypes.data,
                                                  t.ctypes.data, mask.ctypes.data, C.byref(cnt)))
        return q, t, mask[:n].astype(bool), int(cnt.value)
