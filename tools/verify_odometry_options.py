#!/usr/bin/env python3
"""Parity on the ODOMETRY's option set.  Frame2Frame::PNECAlign forces use_nec_ = true, use_ceres_ = false whatever
the YAML says (frame2frame.cc:127-128, quirk C2): what pnec_vo gets per frame pair is the RANSAC eigensolver stage's
pose (pnec.cc:231-281) and its inliers -- no weighted stage, no refinement behind it.  That stage's arithmetic lives in
opengv (not in the reference tree).  The device and the oracle minimise lambda_min(M(R)) with a damped Newton iteration
converged to ~1e-12 rad; opengv's own iteration, as far as its source is remembered ([EXT], unverified), is a normalised
steepest descent that stops once its step length falls below 1e-5, i.e. ~1e-5 rad short of the minimiser.  This tool
puts numbers on the difference: the device's chain with the odometry's options against the oracle's RANSAC eigensolver
run (a) with the Newton iteration (the device's twin) and (b) with that descent (oracle eigensolver scheme 1) in every
eigenvalue minimisation, hypotheses included; and, for contrast, the DEFAULT chain's output (refinement at the end)
against the oracle's default chain run with the descent.   python tools/verify_odometry_options.py [pairs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = 512
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=1, device=dev)
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
with Batch.uniform(capi.MODE_TARGET, P, N) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    vo = capi.default_pipeline_options(use_nec=1, use_ceres=0)
    q_vo, t_vo, m_vo, c_vo = b.solve_pipeline(g.init_q, g.init_t, want_inliers=True, options=vo)
    q_df, t_df, m_df, c_df = b.solve_pipeline(g.init_q, g.init_t, want_inliers=True)
torch.cuda.synchronize()
f1, f2, cv = (x.cpu().numpy() for x in (g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3)))
off = np.arange(P + 1, dtype=np.int64) * N


def angles(qa, qb):
    a, b = np.asarray(qa), np.asarray(qb)
    d = np.abs(np.sum(a * b, axis=1)).clip(0, 1)
    v = np.stack([a[:, 3] * b[:, 0] - a[:, 0] * b[:, 3] - a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1],
                  a[:, 3] * b[:, 1] + a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] - a[:, 2] * b[:, 0],
                  a[:, 3] * b[:, 2] - a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] - a[:, 2] * b[:, 3]], 1)
    return 2.0 * np.arctan2(np.linalg.norm(v, axis=1), d)


st = lambda x: {"max": float(np.max(x)), "p99": float(np.percentile(x, 99)), "median": float(np.median(x)),
                "pairs_over_1e-6_rad": int((x > 1e-6).sum())}
out = {"pairs": P, "corr": N, "outliers": 0.10, "oracle_threads": po.max_threads(),
       "what": "device chain with the odometry's forced options (use_nec, no refinement: the RANSAC eigensolver stage's pose is the output) "
               "and with the default options, against the CPU oracle with its eigenvalue minimisations run by (scheme 0) the damped Newton "
               "iteration the device also runs and (scheme 1) an opengv-style normalised steepest descent that stops at step < 1e-5 "
               "[EXT: restated from memory, unverified]"}
gm = m_vo.cpu().numpy().reshape(P, N).astype(bool)
for scheme, name in ((0, "newton (the device's twin)"), (1, "opengv-style descent [EXT]")):
    po.set_eigensolver_scheme(scheme)
    o = po.solve_chain_batch(off, f1, f2, cv, g.init_q.cpu().numpy(), seed=1, num_threads=po.max_threads())
    om = o["mask"].reshape(P, N)
    same = (om == gm).all(axis=1)
    out[f"scheme_{scheme}"] = {
        "eigenvalue_minimisation": name,
        "odometry_options_rotation_diff_rad": st(angles(q_vo.cpu().numpy(), o["es_q"])),
        "odometry_options_inlier_masks_identical": int(same.sum()),
        "odometry_options_inlier_count_max_abs_diff": int(np.abs(c_vo.cpu().numpy() - o["inlier_count"]).max()),
        "odometry_options_rotation_diff_rad_pairs_with_identical_masks": st(angles(q_vo.cpu().numpy(), o["es_q"])[same]) if same.any() else None,
        "default_options_rotation_diff_rad_after_refinement": st(angles(q_df.cpu().numpy(), o["q"])),
    }
po.set_eigensolver_scheme(0)
print(json.dumps(out))
