"""Batch front-end over the C ABI: a batch of frame pairs resident in HBM and its solves.

numpy arguments go through the HOST memory space of the ABI (blocking, PCIe-inclusive);
torch CUDA tensors go through the DEVICE space (asynchronous on torch's current stream).
torch is plumbing here (device memory, streams); all arithmetic is in libpnec_hip.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import capi


def _is_torch(x) -> bool:
    return x is not None and type(x).__module__.startswith("torch")


@dataclass
class SolveResult:
    """Per solve (pair-major, hypothesis-minor): what PNECCeres::Result() returns + summary."""
    q: object           # [S,4] xyzw, normalised
    t: object           # [S,3] unit
    cost: object        # [S]   1/2 sum r^2 at the returned point
    iterations: object  # [S]   int32
    status: object      # [S]   int32, capi.TERM_NAMES

    def rotation_matrices(self):
        """[S,3,3] from q (numpy or torch, matching the stored arrays)."""
        q = self.q
        x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        xp = __import__("torch") if _is_torch(q) else np
        R = xp.stack([
            xp.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
            xp.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
            xp.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1),
        ], -2)
        return R


class Batch:
    """A batch of independent frame pairs in the solver's SoA layout (``pnec_hip_problem``)."""

    def __init__(self, mode: int, offsets, device: int = 0):
        self._lib = capi.lib()
        self.mode = int(mode)
        self.device = int(device)
        self._offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if self._offsets.ndim != 1 or len(self._offsets) < 1:
            raise ValueError("offsets must be a 1-D array of length n_pairs+1")
        self.n_pairs = len(self._offsets) - 1
        h = C.c_void_p()
        capi.check(self._lib.pnec_hip_problem_create(self.device, self.mode, self.n_pairs,
                                                     self._offsets.ctypes.data, C.byref(h)))
        self._h = h

    @property
    def offsets(self) -> np.ndarray:
        """int64 [n_pairs+1]; for a batch made by select() the first access waits for the device."""
        if self._offsets is None:
            off = np.empty(self.n_pairs + 1, dtype=np.int64)
            capi.check(self._lib.pnec_hip_problem_offsets(self._h, off.ctypes.data))
            self._offsets = off
        return self._offsets

    @classmethod
    def uniform(cls, mode: int, n_pairs: int, n_corr: int, device: int = 0) -> "Batch":
        return cls(mode, np.arange(n_pairs + 1, dtype=np.int64) * n_corr, device)

    @classmethod
    def with_capacity(cls, mode: int, max_pairs: int, max_corr: int, device: int = 0) -> "Batch":
        """A batch that is shaped again and again without allocating (pnec_hip_problem_create_capacity): room for up to
        max_pairs pairs / max_corr correspondences in total, created empty; give it a shape with reshape(), then fill."""
        b = cls.__new__(cls)
        b._lib = capi.lib()
        b.mode, b.device = int(mode), int(device)
        h = C.c_void_p()
        capi.check(b._lib.pnec_hip_problem_create_capacity(b.device, b.mode, int(max_pairs), int(max_corr), C.byref(h)))
        b._h = h
        b.n_pairs = 0
        b._offsets = np.zeros(1, dtype=np.int64)
        return b

    def reshape(self, offsets) -> "Batch":
        """A new shape for a capacity batch (pnec_hip_problem_reshape), asynchronous on torch's current stream: work queued
        earlier still sees the old shape; the planes hold garbage until filled."""
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        stream = None
        try:
            import torch
            if torch.cuda.is_available():
                stream = torch.cuda.current_stream(self.device).cuda_stream
        except ImportError:
            pass
        capi.check(self._lib.pnec_hip_problem_reshape(self._h, len(off) - 1, off.ctypes.data, stream))
        self._offsets = off
        self.n_pairs = len(off) - 1
        return self

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):     # (a select(view=True) result belongs to its source)
                self._lib.pnec_hip_problem_destroy(self._h)
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

    # -- info -------------------------------------------------------------------------------
    @property
    def num_correspondences(self) -> int:
        return self._lib.pnec_hip_problem_num_correspondences(self._h)

    @property
    def max_correspondences(self) -> int:
        return self._lib.pnec_hip_problem_max_correspondences(self._h)

    @property
    def payload_bytes(self) -> int:
        return self._lib.pnec_hip_problem_payload_bytes(self._h)

    def describe_launch(self, options: capi.Options | None = None) -> dict:
        cpl, wpp, ldsk, tpb, res = (C.c_int32() for _ in range(5))
        capi.check(self._lib.pnec_hip_describe_launch(
            self._h, C.byref(options) if options is not None else None, C.byref(cpl),
            C.byref(wpp), C.byref(ldsk), C.byref(tpb), C.byref(res)))
        return {"corr_per_lane": cpl.value, "waves_per_pair": wpp.value,
                "lds_corr_per_lane": ldsk.value, "threads_per_block": tpb.value,
                "resident": bool(res.value)}

    # -- argument hygiene (DEVICE space: the ABI reinterprets the pointer as float64 on self.device)
    def _dev_tensor(self, a, name, shape=None, dtype=None):
        """A contiguous CUDA tensor of `dtype` (float64 by default) on this batch's GPU, or an error."""
        import torch
        if a is None:
            return None
        dtype = dtype or torch.float64
        if not _is_torch(a):
            raise TypeError(f"{name}: mixing numpy and torch arguments in one call is not supported")
        if not a.is_cuda:
            raise ValueError(f"{name}: torch arguments must be CUDA tensors (or pass numpy arrays)")
        if a.device.index != self.device:
            raise ValueError(f"{name}: tensor lives on cuda:{a.device.index}, the batch on cuda:{self.device}")
        if a.dtype != dtype:
            if dtype != torch.float64 or not a.dtype.is_floating_point:
                raise TypeError(f"{name}: expected {dtype}, got {a.dtype}")
            a = a.to(dtype)
        a = a.contiguous()
        if shape is not None and tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
        return a

    def _dev_out(self, a, name, shape, dtype):
        """A caller-supplied output buffer: must already be exactly what the kernel writes."""
        if not (_is_torch(a) and a.is_cuda and a.device.index == self.device and a.dtype == dtype
                and a.is_contiguous() and tuple(a.shape) == tuple(shape)):
            raise ValueError(f"out.{name}: need a contiguous {dtype} CUDA tensor of shape {tuple(shape)} "
                             f"on cuda:{self.device}")
        return a

    # -- ingest -----------------------------------------------------------------------------
    def fill(self, bvs1, bvs2, covs=None, covs_host=None, first_pair: int = 0,
             n_pairs: int | None = None):
        """Fill pairs [first_pair, first_pair+n_pairs) from reference-layout arrays.

        bvs*: [M,3]; covs*: [M,3,3] (symmetric) or [M,9] Eigen column-major; M = the number of
        correspondences in that pair range.  numpy -> HOST space, torch.cuda -> DEVICE space.
        """
        if n_pairs is None:
            n_pairs = self.n_pairs - first_pair
        m = int(self.offsets[first_pair + n_pairs] - self.offsets[first_pair])
        arrays = [bvs1, bvs2, covs, covs_host]
        on_device = _is_torch(bvs1)
        ptrs, keep = [], []
        for i, a in enumerate(arrays):
            if a is None:
                ptrs.append(None)
                continue
            width = 3 if i < 2 else 9
            if on_device:
                import torch
                if not _is_torch(a) or not a.is_cuda:
                    raise ValueError("torch inputs must be CUDA tensors (or pass numpy)")
                if a.device.index != self.device:
                    raise ValueError(f"array {i}: tensor on cuda:{a.device.index}, batch on cuda:{self.device}")
                if a.dim() == 3:  # [M,3,3] symmetric == its own column-major image
                    a = a.reshape(a.shape[0], 9)
                a = a.contiguous().to(torch.float64)
                if a.numel() != m * width:
                    raise ValueError(f"array {i}: expected {m}x{width} values, got {a.numel()}")
                keep.append(a)
                ptrs.append(a.data_ptr())
            else:
                a = np.asarray(a, dtype=np.float64)
                if a.ndim == 3:
                    a = np.transpose(a, (0, 2, 1)).reshape(a.shape[0], 9)
                a = np.ascontiguousarray(a)
                if a.size != m * width:
                    raise ValueError(f"array {i}: expected {m}x{width} values, got {a.size}")
                keep.append(a)
                ptrs.append(a.ctypes.data)
        stream = None
        if on_device:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        capi.check(self._lib.pnec_hip_problem_fill(
            self._h, first_pair, n_pairs, ptrs[0], ptrs[1], ptrs[2], ptrs[3],
            capi.MEM_DEVICE if on_device else capi.MEM_HOST, stream))
        return self

    def fill_keypoints(self, pts1, pts2, cov2=None, cov1=None, K_inv=None, kappa: float = 1.0,
                       first_pair: int = 0, n_pairs: int | None = None):
        """Fused ingest from keypoints (KeyPoint::Unproject, keypoints.cc:49-62): pixel positions [M,2] of
        both frames + image-plane covariances [M,2,2] (or [M,3] = xx, xy, yy) -> bearings and bearing
        covariances computed on the device straight into the SoA planes.  numpy -> HOST space,
        torch.cuda -> DEVICE space."""
        if n_pairs is None:
            n_pairs = self.n_pairs - first_pair
        m = int(self.offsets[first_pair + n_pairs] - self.offsets[first_pair])
        on_device = _is_torch(pts1)
        xp = __import__("torch") if on_device else np

        def c3(a):
            if a is None:
                return None
            if a.ndim == 3:  # [M,2,2] -> (xx, xy, yy)
                a = xp.stack([a[:, 0, 0], a[:, 1, 0], a[:, 1, 1]], -1)
            return a
        K = np.eye(3) if K_inv is None else K_inv
        arrays = [(pts1, 2), (pts2, 2), (c3(cov2), 3), (c3(cov1), 3)]
        ptrs, keep = [], []
        for a, width in arrays:
            if a is None:
                ptrs.append(None)
                continue
            if on_device:
                a = self._dev_tensor(a, "keypoint array")
                ptrs.append(a.data_ptr())
            else:
                a = np.ascontiguousarray(a, dtype=np.float64)
                ptrs.append(a.ctypes.data)
            if int(np.prod(a.shape)) != m * width:
                raise ValueError(f"expected {m}x{width} values, got {tuple(a.shape)}")
            keep.append(a)
        if on_device:
            import torch
            Kd = torch.as_tensor(np.asarray(K.cpu() if _is_torch(K) else K, dtype=np.float64).T.copy().reshape(9),
                                 device=f"cuda:{self.device}")
            kp, stream, space = Kd.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream, capi.MEM_DEVICE
        else:
            Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).T.reshape(9))
            kp, stream, space = Kd.ctypes.data, None, capi.MEM_HOST
        capi.check(self._lib.pnec_hip_problem_fill_keypoints(self._h, first_pair, n_pairs, ptrs[0], ptrs[1], ptrs[2],
                                                             ptrs[3], kp, float(kappa), 1, space, stream))
        return self

    def export_payload(self) -> np.ndarray:
        """The SoA planes as they sit in HBM (float64 [payload_doubles]); for tests."""
        out = np.empty(self._lib.pnec_hip_problem_payload_doubles(self._h))
        capi.check(self._lib.pnec_hip_problem_export_payload(self._h, out.ctypes.data, capi.MEM_HOST, None))
        return out

    # -- solve ------------------------------------------------------------------------------
    def solve(self, init_q, init_t=None, reg: float = 1e-13,
              options: capi.Options | None = None, hyp_t=None, n_hyp: int = 1,
              out: SolveResult | None = None) -> SolveResult:
        """InitValues + Optimize + Result for every (pair, hypothesis) on the device."""
        on_device = _is_torch(init_q)
        n_hyp = int(n_hyp) if hyp_t is not None else 1
        if n_hyp < 1:
            raise ValueError("n_hyp must be >= 1")
        S = self.n_pairs * n_hyp
        if on_device:
            import torch
            init_q = self._dev_tensor(init_q, "init_q", (self.n_pairs, 4))
            init_t = self._dev_tensor(init_t, "init_t", (self.n_pairs, 3))
            hyp_t = self._dev_tensor(hyp_t, "hyp_t", (S, 3))
            dev = init_q.device
            f64 = dict(dtype=torch.float64, device=dev)
            if out is None:
                out = SolveResult(torch.empty((S, 4), **f64), torch.empty((S, 3), **f64),
                                  torch.empty((S,), **f64),
                                  torch.empty((S,), dtype=torch.int32, device=dev),
                                  torch.empty((S,), dtype=torch.int32, device=dev))
            else:
                self._dev_out(out.q, "q", (S, 4), torch.float64)
                self._dev_out(out.t, "t", (S, 3), torch.float64)
                self._dev_out(out.cost, "cost", (S,), torch.float64)
                self._dev_out(out.iterations, "iterations", (S,), torch.int32)
                self._dev_out(out.status, "status", (S,), torch.int32)
            p = lambda a: None if a is None else a.data_ptr()
            stream = torch.cuda.current_stream(self.device).cuda_stream
            space = capi.MEM_DEVICE
        else:
            init_q = np.ascontiguousarray(init_q, dtype=np.float64)
            init_t = None if init_t is None else np.ascontiguousarray(init_t, dtype=np.float64)
            hyp_t = None if hyp_t is None else np.ascontiguousarray(hyp_t, dtype=np.float64)
            if out is None:
                out = SolveResult(np.empty((S, 4)), np.empty((S, 3)), np.empty(S),
                                  np.empty(S, dtype=np.int32), np.empty(S, dtype=np.int32))
            else:
                for name, shape, dt in (("q", (S, 4), np.float64), ("t", (S, 3), np.float64), ("cost", (S,), np.float64),
                                        ("iterations", (S,), np.int32), ("status", (S,), np.int32)):
                    a = getattr(out, name)
                    if not (isinstance(a, np.ndarray) and a.dtype == dt and a.shape == shape and a.flags.c_contiguous):
                        raise ValueError(f"out.{name}: need a C-contiguous {np.dtype(dt).name} array of shape {shape}")
            p = lambda a: None if a is None else a.ctypes.data
            stream = None
            space = capi.MEM_HOST
        if tuple(init_q.shape) != (self.n_pairs, 4):
            raise ValueError("init_q must be [n_pairs,4] (xyzw)")
        if init_t is not None and tuple(init_t.shape) != (self.n_pairs, 3):
            raise ValueError("init_t must be [n_pairs,3]")
        if hyp_t is not None and tuple(hyp_t.shape) != (S, 3):
            raise ValueError("hyp_t must be [n_pairs*n_hyp,3]")
        capi.check(self._lib.pnec_hip_solve(
            self._h, p(init_q), p(init_t), int(n_hyp), p(hyp_t), float(reg),
            C.byref(options) if options is not None else None, p(out.q), p(out.t), p(out.cost),
            p(out.iterations), p(out.status), space, stream))
        return out

    def _front(self, weighted, init_q, init_t, reg, weighted_iterations):
        on_device = _is_torch(init_q)
        if on_device:
            import torch
            keep = [self._dev_tensor(init_q, "init_q", (self.n_pairs, 4)),
                    self._dev_tensor(init_t, "init_t", (self.n_pairs, 3))]
            f64 = dict(dtype=torch.float64, device=keep[0].device)
            oq, ot = torch.empty((self.n_pairs, 4), **f64), torch.empty((self.n_pairs, 3), **f64)
            stream, space = torch.cuda.current_stream(self.device).cuda_stream, capi.MEM_DEVICE
            pq, pt = keep[0].data_ptr(), (None if keep[1] is None else keep[1].data_ptr())
        else:
            keep = [np.ascontiguousarray(init_q, dtype=np.float64),
                    None if init_t is None else np.ascontiguousarray(init_t, dtype=np.float64)]
            oq, ot = np.empty((self.n_pairs, 4)), np.empty((self.n_pairs, 3))
            stream, space = None, capi.MEM_HOST
            pq, pt = keep[0].ctypes.data, (None if keep[1] is None else keep[1].ctypes.data)
        po, pto = (oq.data_ptr(), ot.data_ptr()) if on_device else (oq.ctypes.data, ot.ctypes.data)
        if weighted:
            capi.check(self._lib.pnec_hip_weighted_eigensolver(self._h, pq, pt, float(reg),
                                                              int(weighted_iterations), po, pto, space, stream))
        else:
            capi.check(self._lib.pnec_hip_nec_eigensolver(self._h, pq, po, pto, space, stream))
        return oq, ot

    def ransac_eigensolver(self, init_q, seed: int = 1, max_iterations: int = 5000, sample_size: int = 10,
                           threshold: float = 1e-6):
        """PNEC::Eigensolver with RANSAC (pnec.cc:239-272): -> (q, t, inlier_mask [sumN] uint8,
        inlier_count [P], ransac_iterations [P])"""
        M, P = self.num_correspondences, self.n_pairs
        if _is_torch(init_q):
            import torch
            init_q = self._dev_tensor(init_q, "init_q", (P, 4))
            dev = init_q.device
            q = torch.empty((P, 4), dtype=torch.float64, device=dev)
            t = torch.empty((P, 3), dtype=torch.float64, device=dev)
            mask = torch.empty((max(M, 1),), dtype=torch.uint8, device=dev)
            cnt = torch.empty((P,), dtype=torch.int32, device=dev)
            its = torch.empty((P,), dtype=torch.int32, device=dev)
            iq = init_q.contiguous()
            capi.check(self._lib.pnec_hip_ransac_eigensolver(
                self._h, iq.data_ptr(), int(seed), int(max_iterations), int(sample_size), float(threshold),
                q.data_ptr(), t.data_ptr(), mask.data_ptr(), cnt.data_ptr(), its.data_ptr(), capi.MEM_DEVICE,
                torch.cuda.current_stream(self.device).cuda_stream))
            return q, t, mask[:M], cnt, its
        iq = np.ascontiguousarray(init_q, dtype=np.float64)
        q, t = np.empty((P, 4)), np.empty((P, 3))
        mask = np.zeros(max(M, 1), dtype=np.uint8)
        cnt, its = np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32)
        capi.check(self._lib.pnec_hip_ransac_eigensolver(
            self._h, iq.ctypes.data, int(seed), int(max_iterations), int(sample_size), float(threshold),
            q.ctypes.data, t.ctypes.data, mask.ctypes.data, cnt.ctypes.data, its.ctypes.data, capi.MEM_HOST, None))
        return q, t, mask[:M], cnt, its

    def launch_order_hint(self, enable: bool = True) -> None:
        """Opt in to pnec_hip_problem_launch_order_hint: the next RANSAC launch on this batch dispatches the pairs that
        needed more than one round of hypotheses in the last one first (scheduling only; results do not change)."""
        capi.check(self._lib.pnec_hip_problem_launch_order_hint(self._h, 1 if enable else 0))

    def set_eigensolver_scheme(self, scheme: int) -> None:
        """Which iteration the eigenvalue minimisations of the STAGE calls on this batch run (capi.ES_NEWTON / ES_DESCENT /
        ES_LM; include/pnec_hip.h pnec_hip_eigensolver_scheme).  solve_pipeline takes its own from the options."""
        capi.check(self._lib.pnec_hip_problem_set_eigensolver_scheme(self._h, int(scheme)))

    def set_ransac_flags(self, flags: int) -> None:
        """PNEC_HIP_RANSAC_* bits for the STAGE call ransac_eigensolver on this batch (capi.RANSAC_CHAINED_STARTS: every
        hypothesis starts from the last scored model's rotation, opengv's adapter side effect [EXT]; one hypothesis per
        round).  solve_pipeline takes its own from the options' ransac_flags."""
        capi.check(self._lib.pnec_hip_problem_set_ransac_flags(self._h, int(flags)))

    def select(self, mask, view: bool = False) -> "Batch":
        """PNEC::InlierExtraction (pnec.cc:210-229): new Batch with the masked correspondences.
        view=True: into this batch's cached target (pnec_hip_problem_select_view: nothing allocated after the first
        call, the result belongs to this batch and is replaced by the next such call)."""
        h = C.c_void_p()
        M = self.num_correspondences
        fn = self._lib.pnec_hip_problem_select_view if view else self._lib.pnec_hip_problem_select
        if _is_torch(mask):
            import torch
            m = mask if mask.dtype == torch.uint8 else (mask != 0).to(torch.uint8)
            m = self._dev_tensor(m, "mask", (M,), dtype=torch.uint8)
            capi.check(fn(self._h, m.data_ptr(), capi.MEM_DEVICE, torch.cuda.current_stream(self.device).cuda_stream,
                          C.byref(h)))
        else:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            if m.shape != (M,):
                raise ValueError(f"mask: expected shape ({M},), got {m.shape}")
            capi.check(fn(self._h, m.ctypes.data, capi.MEM_HOST, None, C.byref(h)))
        out = Batch.__new__(Batch)
        out._borrowed = bool(view)
        out._lib, out.mode, out.device, out._h = self._lib, self.mode, self.device, h
        if view:
            # the view lives in this batch's cached target: keep the source alive as long as the view is, and retire
            # the previous view object (the library has just re-pointed the handle it held)
            out._source = self
            old = getattr(self, "_last_view", None)
            old = old() if old is not None else None
            if old is not None:
                old._h = None
            import weakref
            self._last_view = weakref.ref(out)
        out.n_pairs = self.n_pairs
        # InlierExtraction ran on the device and nothing was read back: the new batch's offsets are
        # fetched from the library on first use (that call waits for the stream)
        out._offsets = None
        return out

    def solve_pipeline(self, init_q, init_t, options: capi.PipelineOptions | None = None, want_inliers: bool = False):
        """PNEC::Solve (pnec.cc:77-124) for every pair, all stages on the device without a host round
        trip: -> (q [P,4], t [P,3]) or, with want_inliers, (q, t, inlier_mask [sumN] uint8, inlier_count [P])."""
        P = self.n_pairs
        opt = C.byref(options) if options is not None else None
        if _is_torch(init_q):
            import torch
            iq = self._dev_tensor(init_q, "init_q", (P, 4))
            it = self._dev_tensor(init_t, "init_t", (P, 3))
            q = torch.empty((P, 4), dtype=torch.float64, device=iq.device)
            t = torch.empty((P, 3), dtype=torch.float64, device=iq.device)
            mask = cnt = None
            if want_inliers:
                mask = torch.empty((max(self.num_correspondences, 1),), dtype=torch.uint8, device=iq.device)
                cnt = torch.empty((P,), dtype=torch.int32, device=iq.device)
            capi.check(self._lib.pnec_hip_solve_pipeline(
                self._h, iq.data_ptr(), it.data_ptr(), opt, q.data_ptr(), t.data_ptr(),
                None if mask is None else mask.data_ptr(), None if cnt is None else cnt.data_ptr(), capi.MEM_DEVICE,
                torch.cuda.current_stream(self.device).cuda_stream))
            return (q, t, mask[:self.num_correspondences], cnt) if want_inliers else (q, t)
        iq = np.ascontiguousarray(init_q, dtype=np.float64)
        it = np.ascontiguousarray(init_t, dtype=np.float64)
        if iq.shape != (P, 4) or it.shape != (P, 3):
            raise ValueError("init_q must be [n_pairs,4], init_t [n_pairs,3]")
        q, t = np.empty((P, 4)), np.empty((P, 3))
        mask = np.zeros(max(self.num_correspondences, 1), dtype=np.uint8) if want_inliers else None
        cnt = np.zeros(P, dtype=np.int32) if want_inliers else None
        capi.check(self._lib.pnec_hip_solve_pipeline(
            self._h, iq.ctypes.data, it.ctypes.data, opt, q.ctypes.data, t.ctypes.data,
            None if mask is None else mask.ctypes.data, None if cnt is None else cnt.ctypes.data, capi.MEM_HOST, None))
        return (q, t, mask[:self.num_correspondences], cnt) if want_inliers else (q, t)

    def nec_eigensolver(self, init_q):
        """PNEC::Eigensolver without RANSAC (pnec.cc:273-278): -> (q [P,4], t [P,3])"""
        return self._front(False, init_q, None, 0.0, 0)

    def weighted_eigensolver(self, init_q, init_t, reg: float = 1e-13, weighted_iterations: int = 10):
        """PNEC::WeightedEigensolver (pnec.cc:283-348): -> (q [P,4], t [P,3])"""
        return self._front(True, init_q, init_t, reg, weighted_iterations)

    def cost_function(self, q, t):
        """pnec::common::CostFunction per pair (TARGET-mode batches)."""
        if _is_torch(q):
            import torch
            q = self._dev_tensor(q, "q", (self.n_pairs, 4))
            t = self._dev_tensor(t, "t", (self.n_pairs, 3))
            out = torch.empty((self.n_pairs,), dtype=torch.float64, device=q.device)
            capi.check(self._lib.pnec_hip_cost_function(
                self._h, q.data_ptr(), t.data_ptr(), out.data_ptr(),
                capi.MEM_DEVICE, torch.cuda.current_stream(self.device).cuda_stream))
            return out
        q = np.ascontiguousarray(q, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        out = np.empty(self.n_pairs)
        capi.check(self._lib.pnec_hip_cost_function(self._h, q.ctypes.data, t.ctypes.data,
                                                    out.ctypes.data, capi.MEM_HOST, None))
        return out


def select_best(cost, n_hyp: int, device: int = 0):
    """Index of the lowest-cost hypothesis per pair."""
    L = capi.lib()
    if _is_torch(cost):
        import torch
        n_pairs = cost.numel() // n_hyp
        best = torch.empty((n_pairs,), dtype=torch.int32, device=cost.device)
        capi.check(L.pnec_hip_select_best(n_pairs, n_hyp, cost.contiguous().data_ptr(),
                                          best.data_ptr(), capi.MEM_DEVICE, device,
                                          torch.cuda.current_stream(device).cuda_stream))
        return best
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    n_pairs = cost.size // n_hyp
    best = np.empty(n_pairs, dtype=np.int32)
    capi.check(L.pnec_hip_select_best(n_pairs, n_hyp, cost.ctypes.data, best.ctypes.data,
                                      capi.MEM_HOST, device, None))
    return best
