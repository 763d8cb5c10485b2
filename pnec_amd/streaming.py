"""Per-frame (streaming) front-end over ``pnec_hip_stream_*``: what the reference's odometry does --
one ``PNECCeres::Optimize`` / ``PNEC::Solve`` call per frame pair (frame2frame.cc:122-141) -- without a
batch object per call.  A ``Stream`` owns pinned staging slots and a HIP stream; ``submit`` copies the
frame pair in and launches one kernel, ``wait`` polls a flag the kernel raises.  numpy in, numpy out.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .batch import SolveResult


class Stream:
    def __init__(self, max_corr: int = 4096, max_pairs: int = 1, slots: int = 8, device: int = 0):
        self._lib = capi.lib()
        self.max_corr, self.max_pairs, self.slots, self.device = int(max_corr), int(max_pairs), int(slots), int(device)
        h = C.c_void_p()
        capi.check(self._lib.pnec_hip_stream_create(self.device, self.max_corr, self.max_pairs, self.slots, None,
                                                    C.byref(h)))
        self._h = h
        self._pairs = {}   # ticket -> pairs it carries (sizes the result arrays of wait)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pnec_hip_stream_destroy(self._h)
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

    def submit(self, mode: int, bvs1, bvs2, covs=None, covs_host=None, init_q=None, init_t=None, reg: float = 1e-13,
               options: capi.Options | None = None, offsets=None) -> int:
        """One frame pair (or, with `offsets`, several) -> ticket.  Arrays as in Batch.fill; a start pose
        (init_q xyzw, init_t) per pair is required."""
        if init_q is None or init_t is None:
            raise ValueError("init_q and init_t are required: one start pose per pair")
        b1 = np.ascontiguousarray(bvs1, dtype=np.float64).reshape(-1, 3)
        b2 = np.ascontiguousarray(bvs2, dtype=np.float64).reshape(-1, 3)
        M = len(b1)
        off = np.array([0, M], dtype=np.int64) if offsets is None else np.ascontiguousarray(offsets, dtype=np.int64)
        P = len(off) - 1
        if off[-1] != M or len(b2) != M:
            raise ValueError("bvs1, bvs2 and offsets disagree")

        def cov9(a):
            if a is None:
                return None
            a = np.asarray(a, dtype=np.float64)
            if a.ndim == 3:  # [M,3,3] -> Eigen column-major
                a = np.transpose(a, (0, 2, 1))
            a = np.ascontiguousarray(a.reshape(-1, 9))
            if len(a) != M:
                raise ValueError("covariances and bearings disagree in length")
            return a
        c2, c1 = cov9(covs), cov9(covs_host)
        q = np.ascontiguousarray(init_q, dtype=np.float64).reshape(-1, 4)
        t = np.ascontiguousarray(init_t, dtype=np.float64).reshape(-1, 3)
        if len(q) != P or len(t) != P:
            raise ValueError("one start pose per pair")
        p = lambda a: None if a is None else a.ctypes.data
        ticket = C.c_int64(0)
        capi.check(self._lib.pnec_hip_stream_submit(self._h, int(mode), P, off.ctypes.data, p(b1), p(b2), p(c2), p(c1),
                                                    q.ctypes.data, t.ctypes.data, float(reg),
                                                    C.byref(options) if options is not None else None,
                                                    C.byref(ticket)))
        self._pairs[ticket.value] = P
        return ticket.value

    def poll(self, ticket: int) -> bool:
        done = C.c_int32(0)
        capi.check(self._lib.pnec_hip_stream_poll(self._h, int(ticket), C.byref(done)))
        return bool(done.value)

    def wait(self, ticket: int) -> SolveResult:
        P = self._pairs.get(int(ticket))
        if P is None:
            # not a ticket this object handed out (or one already collected).  The library is NOT asked: a wait with
            # NULL outputs on a ticket it does know would consume it and drop its results.
            raise capi.PnecHipError(capi.ERR_INVALID_ARGUMENT, "unknown or already collected ticket")
        del self._pairs[int(ticket)]
        out = SolveResult(np.empty((P, 4)), np.empty((P, 3)), np.empty(P), np.empty(P, dtype=np.int32),
                          np.empty(P, dtype=np.int32))
        capi.check(self._lib.pnec_hip_stream_wait(self._h, int(ticket), out.q.ctypes.data, out.t.ctypes.data,
                                                  out.cost.ctypes.data, out.iterations.ctypes.data,
                                                  out.status.ctypes.data))
        return out

    def solve(self, mode, bvs1, bvs2, covs=None, covs_host=None, init_q=None, init_t=None, reg=1e-13, options=None):
        """submit + wait for one frame pair: PNECCeres::Optimize / NECCeres::Optimize."""
        return self.wait(self.submit(mode, bvs1, bvs2, covs, covs_host, init_q, init_t, reg, options))
