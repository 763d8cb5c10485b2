#!/usr/bin/env python3
"""Parity of the whole PNEC::Solve chain against the oracle's chain on the KITTI-like SYNTHETIC stream (forward
motion, low parallax, ragged sizes, 10 % gross outliers): the data regime of BASELINE configs 3 and 5.
Runs on the GPU box.   python tools/verify_pipeline_kitti.py [P]"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=500, seed=11, device=dev)
M = f1.shape[0]
bad = torch.rand(M, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(M, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
rnd[:, 2] = rnd[:, 2].abs() + 1.0          # in front of the camera, like a wrong match would be
f2 = torch.where(bad[:, None], rnd / rnd.norm(dim=-1, keepdim=True), f2)
off = offsets.cpu().numpy() if hasattr(offsets, "cpu") else np.asarray(offsets)
with Batch(capi.MODE_TARGET, off) as b:
    b.fill(f1, f2, c2)
    q_dev, t_dev, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
    qr, tr, mask_r, cnt_r, its = b.ransac_eigensolver(q0, seed=1)
    torch.cuda.synchronize()
gq, gmask, gits = q_dev.cpu().numpy(), mask.cpu().numpy().astype(bool), its.cpu().numpy()
assert bool(torch.equal(mask, mask_r))
f1n, f2n, c2n, q0n = f1.cpu().numpy(), f2.cpu().numpy(), c2.cpu().numpy(), q0.cpu().numpy()
R0 = np.stack([po.rot_from_quat(q) for q in q0n])


def one(p):
    a, e = off[p], off[p + 1]
    Rr, trr, m, it = po.ransac_eigensolver(f1n[a:e], f2n[a:e], R0[p], seed=1, pair_id=p)
    Rw, tww = po.weighted_eigensolver(f1n[a:e][m], f2n[a:e][m], c2n[a:e][m], Rr, trr)
    s = po.solve(po.MODE_TARGET, f1n[a:e][m], f2n[a:e][m], c2n[a:e][m], None, 1e-13, po.quat_from_rot(Rw), tww, po.default_options())
    return (np.radians(po.rotational_difference_deg(s.R, po.rot_from_quat(gq[p]))), bool((m == gmask[a:e]).all()), int(it) == int(gits[p]), int(m.sum()))


with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as ex:
    out = list(ex.map(one, range(P)))
ang = np.array([o[0] for o in out])
worst = np.argsort(-ang)[:5]
print(json.dumps({"workload": "KITTI-like SYNTHETIC stream (forward motion), ragged, 10 % gross outliers; whole chain in ONE call (pnec_hip_solve_pipeline)",
                  "pairs": P, "corr_min_mean_max": [int(np.diff(off).min()), float(np.diff(off).mean()), int(np.diff(off).max())],
                  "mean_inliers": float(np.mean([o[3] for o in out])), "mean_ransac_iterations": float(gits.mean()),
                  "max_rot_diff_rad": float(ang.max()), "p99_rot_diff_rad": float(np.percentile(ang, 99)), "median_rot_diff_rad": float(np.median(ang)),
                  "pairs_over_1e-6_rad": int((ang > 1e-6).sum()), "inlier_masks_identical": int(sum(o[1] for o in out)),
                  "ransac_iteration_counts_identical": int(sum(o[2] for o in out)),
                  "worst_pairs": [{"pair": int(p), "rot_diff_rad": float(ang[p]), "masks_identical": out[p][1]} for p in worst]}))
