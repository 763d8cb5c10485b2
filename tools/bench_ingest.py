#!/usr/bin/env python3
"""PCIe-inclusive ingest: host arrays -> SoA planes in HBM, the ready-made route (bearings + 3x3
covariances, 120 B per correspondence: pnec_hip_problem_fill) against the fused keypoint route (pixel
positions + 2x2 covariances, 56 B: pnec_hip_problem_fill_keypoints, Unproject + UnscentedTransform on the
device).  Host arrays are pinned (torch pin_memory) so the copies run at link speed.  One JSON object.
  python tools/bench_ingest.py [pairs] [corr]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pnec_amd import Batch, capi, frontend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
M = B * N
rng = np.random.default_rng(1)
K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
Kinv = np.linalg.inv(K)
pin = lambda a: torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()
p1 = pin(np.stack([rng.uniform(0, 1241, M), rng.uniform(0, 376, M)], 1))
p2 = pin(p1 + rng.normal(size=(M, 2)) * 5)
A = rng.normal(size=(M, 2, 2)) * 0.4
c2 = A @ np.transpose(A, (0, 2, 1)) + 0.02 * np.eye(2)
c2x3 = pin(np.stack([c2[:, 0, 0], c2[:, 1, 0], c2[:, 1, 1]], 1))
# the ready-made arrays (what the CPU front end of the reference would hand over)
mu = np.concatenate([p2, np.ones((M, 1))], 1)
b2, S2 = frontend.unscented_transform(mu, np.pad(c2, ((0, 0), (0, 1), (0, 1))), Kinv, 1.0, frontend.CAMERA_PINHOLE)
b1, _ = frontend.unscented_transform(np.concatenate([p1, np.ones((M, 1))], 1), np.pad(c2, ((0, 0), (0, 1), (0, 1))), Kinv, 1.0,
                                     frontend.CAMERA_PINHOLE)
b1 = pin(b1)
b2 = pin(b2)
S9 = pin(np.transpose(S2, (0, 2, 1)).reshape(M, 9))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return float(np.median(ts))


with Batch.uniform(capi.MODE_TARGET, B, N) as a, Batch.uniform(capi.MODE_TARGET, B, N) as b:
    t_fill = timed(lambda: a.fill(b1, b2, S9))
    t_kp = timed(lambda: b.fill_keypoints(p1, p2, c2x3, K_inv=Kinv))
    same = bool(np.array_equal(a.export_payload(), b.export_payload()))
    # device-side cost alone (inputs already in HBM)
    d = lambda x: torch.from_numpy(x).cuda()
    db1, db2, dS9, dp1, dp2, dc = d(b1), d(b2), d(S9), d(p1), d(p2), d(c2x3)
    t_fill_dev = timed(lambda: a.fill(db1, db2, dS9))
    t_kp_dev = timed(lambda: b.fill_keypoints(dp1, dp2, dc, K_inv=Kinv))
print(json.dumps({
    "workload": f"{B} pairs x {N} correspondences, pinned host arrays -> SoA planes in HBM",
    "fill (bearings + 3x3 covariances, 120 B/corr)": {"ms": t_fill * 1e3, "pairs_per_s": B / t_fill, "GB_per_s_over_the_bus": M * 120 / t_fill / 1e9,
                                                       "device_only_ms": t_fill_dev * 1e3},
    "fill_keypoints (pixels + 2x2 covariances, 56 B/corr)": {"ms": t_kp * 1e3, "pairs_per_s": B / t_kp, "GB_per_s_over_the_bus": M * 56 / t_kp / 1e9,
                                                              "device_only_ms": t_kp_dev * 1e3},
    "planes_bitwise_identical": same}), flush=True)
