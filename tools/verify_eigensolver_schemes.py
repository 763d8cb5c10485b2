#!/usr/bin/env python3
"""Parity of the eigensolver stage under every eigensolver scheme, and what the choice of scheme is worth.

opengv::relative_pose::eigensolver (pnec.cc:239-258, :274, :315) is not in the reference tree; the device and the CPU
checker hold three restatements of its eigenvalue minimisation (include/pnec_hip.h pnec_hip_eigensolver_scheme): 0 damped
Newton, 1 normalised descent [EXT], 2 Eigen's Levenberg-Marquardt on the reduced-Cayley gradient [EXT].  On the
reference's odometry path the eigensolver stage IS the output (frame2frame.cc:127-128 forces use_nec, no refinement).
For 2 000 pairs x 512 correspondences with 10 % gross mismatches this prints, for every (device scheme d, checker scheme c):
  * odometry options: rotation difference of the stage's pose, inlier masks / RANSAC iteration counts identical;
  * default options: rotation difference after the refinement;
  * use_ransac_ = false on the same pairs WITHOUT the mismatches: rotation difference of the plain eigensolver.
The diagonal d == c is the parity statement (device against its sequential twin); the off-diagonal entries say how far
apart the three recollections are -- i.e. what is at stake in the choice.   python tools/verify_eigensolver_schemes.py [pairs] [share of mismatches]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = 512
OUTLIERS = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10   # share of gross mismatches
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=1, device=dev)
clean2 = g.bvs2.clone()
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < OUTLIERS
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
NAMES = {0: "newton", 1: "descent [EXT]", 2: "lm, reduced Cayley [EXT]"}


def angles(qa, qb):
    a, b = np.asarray(qa), np.asarray(qb)
    d = np.abs(np.sum(a * b, axis=1)).clip(0, 1)
    v = np.stack([a[:, 3] * b[:, 0] - a[:, 0] * b[:, 3] - a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1],
                  a[:, 3] * b[:, 1] + a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] - a[:, 2] * b[:, 0],
                  a[:, 3] * b[:, 2] - a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] - a[:, 2] * b[:, 3]], 1)
    return 2.0 * np.arctan2(np.linalg.norm(v, axis=1), d)


st = lambda x: {"max": float(np.max(x)), "p99": float(np.percentile(x, 99)), "median": float(np.median(x)),
                "pairs_over_1e-8_rad": int((x > 1e-8).sum()), "pairs_over_1e-6_rad": int((x > 1e-6).sum())}
devr = {}
with Batch.uniform(capi.MODE_TARGET, P, N) as b, Batch.uniform(capi.MODE_NEC, P, N) as bc:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    bc.fill(g.bvs1.reshape(-1, 3), clean2.reshape(-1, 3))
    for d in (0, 1, 2):
        vo = capi.default_pipeline_options(use_nec=1, use_ceres=0, eigensolver_scheme=d)
        df = capi.default_pipeline_options(eigensolver_scheme=d)
        r = {}
        r["q_vo"], _, r["m_vo"], r["c_vo"] = b.solve_pipeline(g.init_q, g.init_t, want_inliers=True, options=vo)
        r["q_df"], _, _, _ = b.solve_pipeline(g.init_q, g.init_t, want_inliers=True, options=df)
        b.set_eigensolver_scheme(d)
        _, _, _, _, r["its"] = b.ransac_eigensolver(g.init_q, seed=1)
        bc.set_eigensolver_scheme(d)
        r["q_plain"], _ = bc.nec_eigensolver(g.init_q)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.solve_pipeline(g.init_q, g.init_t, options=vo)
        torch.cuda.synchronize()
        r["vo_pairs_per_s"] = 3 * P / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(3):
            b.solve_pipeline(g.init_q, g.init_t, options=df)
        torch.cuda.synchronize()
        r["df_pairs_per_s"] = 3 * P / (time.perf_counter() - t0)
        devr[d] = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in r.items()}
f1, f2, cv = (x.cpu().numpy() for x in (g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3)))
f2c = clean2.reshape(-1, 3).cpu().numpy()
R0 = g.init_R.cpu().numpy()
off = np.arange(P + 1, dtype=np.int64) * N
out = {"pairs": P, "corr": N, "outliers": OUTLIERS, "oracle_threads": po.max_threads(), "schemes": NAMES,
       "what": __doc__.split("\n\n")[1].replace("\n", " "), "device_vs_checker": {}, "device_vs_device": {},
       "device_pairs_per_s_at_this_size": {str(d): {"odometry_options": devr[d]["vo_pairs_per_s"], "default_options": devr[d]["df_pairs_per_s"]}
                                           for d in devr}}
for c in (0, 1, 2):
    po.set_eigensolver_scheme(c)
    o = po.solve_chain_batch(off, f1, f2, cv, g.init_q.cpu().numpy(), seed=1, num_threads=po.max_threads())
    plain = np.zeros((P, 4))
    for p in range(P):
        Ro, _ = po.nec_eigensolver(f1[p * N:(p + 1) * N], f2c[p * N:(p + 1) * N], R0[p])
        plain[p] = po.quat_from_rot(Ro)
    om = o["mask"].reshape(P, N)
    for d in (0, 1, 2):
        r = devr[d]
        same = (om == r["m_vo"].reshape(P, N).astype(bool)).all(axis=1)
        a_vo = angles(r["q_vo"], o["es_q"])
        out["device_vs_checker"][f"device_{d}_checker_{c}"] = {
            "odometry_options_rotation_diff_rad": st(a_vo),
            "odometry_options_rotation_diff_rad_pairs_with_identical_masks": st(a_vo[same]) if same.any() else None,
            "inlier_masks_identical": int(same.sum()),
            "ransac_iteration_counts_identical": int((r["its"] == o["ransac_iterations"]).sum()),
            "default_options_rotation_diff_rad_after_refinement": st(angles(r["q_df"], o["q"])),
            "no_ransac_clean_data_rotation_diff_rad": st(angles(r["q_plain"], plain)),
        }
# the device's scheme 0 against the checker with the round-3 RANSAC rules (every hypothesis scored, 50 iterations): what
# the 25-iteration rule for hypotheses (a cut-off minimisation yields no model) changes
po.set_eigensolver_scheme(0)
po.set_ransac_frozen_rules(True)
o = po.solve_chain_batch(off, f1, f2, cv, g.init_q.cpu().numpy(), seed=1, num_threads=po.max_threads())
po.set_ransac_frozen_rules(False)
r = devr[0]
same = (o["mask"].reshape(P, N) == r["m_vo"].reshape(P, N).astype(bool)).all(axis=1)
a_vo = angles(r["q_vo"], o["es_q"])
out["device_0_vs_checker_0_with_round3_ransac_rules"] = {
    "rules": "every hypothesis scored, a hypothesis' minimisation may take 50 iterations (oracle: pnec_oracle_set_ransac_frozen_rules)",
    "inlier_masks_identical": int(same.sum()),
    "ransac_iteration_counts_identical": int((r["its"] == o["ransac_iterations"]).sum()),
    "odometry_options_rotation_diff_rad": st(a_vo),
    "odometry_options_rotation_diff_rad_pairs_with_identical_masks": st(a_vo[same]) if same.any() else None,
    "default_options_rotation_diff_rad_after_refinement": st(angles(r["q_df"], o["q"])),
}
for d in (1, 2):
    out["device_vs_device"][f"device_{d}_vs_device_0"] = {
        "odometry_options_rotation_diff_rad": st(angles(devr[d]["q_vo"], devr[0]["q_vo"])),
        "inlier_masks_identical": int((devr[d]["m_vo"].reshape(P, N) == devr[0]["m_vo"].reshape(P, N)).all(axis=1).sum()),
        "default_options_rotation_diff_rad_after_refinement": st(angles(devr[d]["q_df"], devr[0]["q_df"])),
        "no_ransac_clean_data_rotation_diff_rad": st(angles(devr[d]["q_plain"], devr[0]["q_plain"])),
    }
print(json.dumps(out))
