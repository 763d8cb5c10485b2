"""Input side of the hot path: covariance propagation from the image plane to bearing space
(`pnec::common::UnscentedTransform` / `Unproject`, src/common/common.cc:460-525) on the device."""
from __future__ import annotations

import numpy as np

from . import capi
from .batch import _is_torch

CAMERA_OMNIDIRECTIONAL, CAMERA_PINHOLE = 0, 1


def unscented_transform(mu, covs, K_inv=None, kappa: float = 1.0, camera_model: int = CAMERA_PINHOLE,
                        device: int = 0):
    """mu [n,3], covs [n,3,3] -> (bearings [n,3], bearing covariances [n,3,3]).

    numpy in -> numpy out (host space); torch.cuda in -> torch.cuda out (device space, async)."""
    L = capi.lib()
    if _is_torch(mu):
        import torch
        n = mu.shape[0]
        mu = mu.contiguous().to(torch.float64)
        c9 = covs.transpose(-1, -2).reshape(n, 9).contiguous().to(torch.float64)  # column-major
        K = torch.eye(3, dtype=torch.float64, device=mu.device) if K_inv is None else K_inv.to(mu.device, torch.float64)
        K9 = K.t().reshape(9).contiguous()
        bvs = torch.empty((n, 3), dtype=torch.float64, device=mu.device)
        out = torch.empty((n, 9), dtype=torch.float64, device=mu.device)
        capi.check(L.pnec_hip_unscented_transform(n, mu.data_ptr(), c9.data_ptr(), K9.data_ptr(), float(kappa),
                                                  int(camera_model), bvs.data_ptr(), out.data_ptr(),
                                                  capi.MEM_DEVICE, device,
                                                  torch.cuda.current_stream(device).cuda_stream))
        return bvs, out.reshape(n, 3, 3).transpose(-1, -2)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    n = mu.shape[0]
    c9 = np.ascontiguousarray(np.transpose(np.asarray(covs, dtype=np.float64), (0, 2, 1)).reshape(n, 9))
    K = np.eye(3) if K_inv is None else np.asarray(K_inv, dtype=np.float64)
    K9 = np.ascontiguousarray(K.T.reshape(9))
    bvs = np.empty((n, 3))
    out = np.empty((n, 9))
    capi.check(L.pnec_hip_unscented_transform(n, mu.ctypes.data, c9.ctypes.data, K9.ctypes.data, float(kappa),
                                              int(camera_model), bvs.ctypes.data, out.ctypes.data,
                                              capi.MEM_HOST, device, None))
    return bvs, np.transpose(out.reshape(n, 3, 3), (0, 2, 1))
