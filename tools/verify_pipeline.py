#!/usr/bin/env python3
"""Parity of the whole PNEC::Solve chain (RANSAC eigensolver -> inlier extraction -> weighted eigensolver + SCF
-> refinement, reference-default Options) against the oracle's chain, pair by pair, on P pairs of the
pipeline benchmark's workload (512 correspondences, 10 % gross outliers).  Runs on the GPU box; the oracle
side is ~11 ms per pair per host thread.   python tools/verify_pipeline.py [P] [seed]"""
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
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 512
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=SEED, device=dev)
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
with Batch.uniform(capi.MODE_TARGET, P, N) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    qr, tr, mask, cnt, its = b.ransac_eigensolver(g.init_q, seed=1)
    sel = b.select(mask)
    qw, tw = sel.weighted_eigensolver(qr, tr, 1e-13, 10)
    res = sel.solve(qw, tw)
    sel.close()
gq = res.q.cpu().numpy()
gqr, gqw, git_ls = qr.cpu().numpy(), qw.cpu().numpy(), res.iterations.cpu().numpy()
gtw = tw.cpu().numpy()
gmask = mask.cpu().numpy().reshape(P, N).astype(bool)
gits = its.cpu().numpy()
f1, f2, c2, R0 = g.bvs1.cpu().numpy(), g.bvs2.cpu().numpy(), g.covs2.cpu().numpy(), g.init_R.cpu().numpy()


def one(p):
    Rr, trr, m, it = po.ransac_eigensolver(f1[p], f2[p], R0[p], seed=1, pair_id=p)
    Rw, tww = po.weighted_eigensolver(f1[p][m], f2[p][m], c2[p][m], Rr, trr)
    s = po.solve(po.MODE_TARGET, f1[p][m], f2[p][m], c2[p][m], None, 1e-13, po.quat_from_rot(Rw), tww, po.default_options())
    return (np.radians(po.rotational_difference_deg(s.R, po.rot_from_quat(gq[p]))), bool((m == gmask[p]).all()), int(it) == int(gits[p]),
            np.radians(po.rotational_difference_deg(Rr, po.rot_from_quat(gqr[p]))),
            np.radians(po.rotational_difference_deg(Rw, po.rot_from_quat(gqw[p]))), int(s.iterations), int(git_ls[p]))


with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as ex:
    out = list(ex.map(one, range(P)))
ang = np.array([o[0] for o in out])
a_r, a_w = np.array([o[3] for o in out]), np.array([o[4] for o in out])
worst = np.argsort(-ang)[:5]
# attribution: where the LS iteration counts differ, run the oracle's LS from the DEVICE's own weighted-stage
# output (identical inputs to the LS stage): if that agrees, the difference came from the inputs' last bits
# meeting a stopping threshold, not from the LS kernel
attrib = []
for p in range(P):
    if out[p][5] != out[p][6]:
        m = gmask[p]
        s2 = po.solve(po.MODE_TARGET, f1[p][m], f2[p][m], c2[p][m], None, 1e-13, gqw[p], gtw[p], po.default_options())
        attrib.append({"pair": int(p), "oracle_ls_iterations_from_device_inputs": int(s2.iterations), "device_ls_iterations": int(git_ls[p]),
                       "rot_diff_rad_given_identical_ls_inputs": float(np.radians(po.rotational_difference_deg(s2.R, po.rot_from_quat(gq[p]))))})
stages = {"ls_stage_on_identical_inputs_where_counts_differed": attrib, "after_ransac_eigensolver_rot_diff_rad": {"max": float(a_r.max()), "p99": float(np.percentile(a_r, 99))},
          "after_weighted_eigensolver_rot_diff_rad": {"max": float(a_w.max()), "p99": float(np.percentile(a_w, 99))},
          "ls_iteration_counts_identical": int(sum(o[5] == o[6] for o in out)),
          "worst_pairs": [{"pair": int(p), "final": float(ang[p]), "after_ransac": float(a_r[p]), "after_weighted": float(a_w[p]),
                           "ls_iterations_oracle_device": [out[p][5], out[p][6]]} for p in worst]}
print(json.dumps({"stages": stages, "pairs": P, "corr": N, "chain": "RANSAC eigensolver -> inlier extraction -> weighted eigensolver + SCF -> refinement",
                  "max_rot_diff_rad": float(ang.max()), "p99_rot_diff_rad": float(np.percentile(ang, 99)),
                  "median_rot_diff_rad": float(np.median(ang)), "pairs_over_1e-6_rad": int((ang > 1e-6).sum()),
                  "inlier_masks_identical": int(sum(o[1] for o in out)), "ransac_iteration_counts_identical": int(sum(o[2] for o in out))}))
