#!/usr/bin/env python3
"""Parity of the WHOLE benchmark batch (bench.py's 100k pairs x 512 correspondences, same seeds)
against the reference-faithful CPU oracle (central-difference Jacobian + Ceres LM policy), chunk by
chunk, in the fixed-10-iteration mode of the bench line and with Ceres-default termination.
Runs on the GPU box (~1 min of host CPU); prints one JSON object per mode."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
N, CH = 512, 10_000
dev = torch.device("cuda:0")


def quat_angle(a, b):
    d = np.clip(np.abs(np.sum(a * b, axis=-1)), 0, 1)
    v = np.linalg.norm(a[..., :3] * b[..., 3:4] - b[..., :3] * a[..., 3:4] - np.cross(a[..., :3], b[..., :3]), axis=-1)
    return 2 * np.arctan2(v, d)


for label, conv in (("fixed 10 LM iterations (the bench line)", 0), ("Ceres-default termination", 1)):
    kw = dict(check_convergence=conv)
    if not conv:
        kw["max_num_iterations"] = 10
    opts = capi.default_options(**kw)
    oo = po.default_options(jacobian_mode=po.JAC_NUMERIC_CENTRAL, **kw)
    ang, it_equal, st_equal, n = [], 0, 0, 0
    for c, first in enumerate(range(0, B, CH)):
        m = min(CH, B - first)
        g = sim.generate(m, N, noise_type="anisotropic_inhomogeneous", noise_level=1.0, seed=1 + c, device=dev)
        with Batch.uniform(capi.MODE_TARGET, m, N) as b:
            b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
            res = b.solve(g.init_q, g.init_t, reg=1e-13, options=opts)
            gq, gi, gs = res.q.cpu().numpy(), res.iterations.cpu().numpy(), res.status.cpu().numpy()
        q, t, cost, it, st = po.solve_batch(
            po.MODE_TARGET, np.arange(m + 1, dtype=np.int64) * N, g.bvs1.reshape(-1, 3).cpu().numpy(),
            g.bvs2.reshape(-1, 3).cpu().numpy(), po.covs_to_colmajor9(g.covs2.reshape(-1, 3, 3).cpu().numpy()), None,
            1e-13, g.init_q.cpu().numpy(), g.init_t.cpu().numpy(), options=oo, num_threads=po.max_threads())
        ang.append(quat_angle(gq, q))
        it_equal += int((gi == it).sum())
        st_equal += int((gs == st).sum())
        n += m
        del g
    ang = np.concatenate(ang)
    print(json.dumps({"mode": label, "pairs": n, "corr": N, "against": "oracle, central-difference Jacobian (the reference's configuration)",
                      "max_rot_diff_rad": float(ang.max()), "p99_rot_diff_rad": float(np.percentile(ang, 99)),
                      "median_rot_diff_rad": float(np.median(ang)), "pairs_over_1e-6_rad": int((ang > 1e-6).sum()),
                      "iteration_counts_equal": it_equal, "termination_codes_equal": st_equal}), flush=True)
