#!/usr/bin/env python3
"""Parity of the WHOLE benchmark batch (bench.py's 100k pairs x 512 correspondences, same seeds)
against the reference-faithful CPU oracle (central-difference Jacobian + Ceres LM policy), chunk by
chunk, in the fixed-10-iteration mode of the bench line and with Ceres-default termination.
Runs on the GPU box (~1 min of host CPU per family); prints one JSON object per family and mode.
   python tools/verify_full_batch.py [B] [nec,target,host,sym]"""
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
FAMILIES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["target"]   # any of nec,target,host,sym
N, CH = 512, 10_000
MODES = {"nec": (capi.MODE_NEC, po.MODE_NEC), "target": (capi.MODE_TARGET, po.MODE_TARGET),
         "host": (capi.MODE_HOST, po.MODE_HOST), "sym": (capi.MODE_SYM, po.MODE_SYM)}
dev = torch.device("cuda:0")


def quat_angle(a, b):
    d = np.clip(np.abs(np.sum(a * b, axis=-1)), 0, 1)
    v = np.linalg.norm(a[..., :3] * b[..., 3:4] - b[..., :3] * a[..., 3:4] - np.cross(a[..., :3], b[..., :3]), axis=-1)
    return 2 * np.arctan2(v, d)


for fam, label, conv in [(f, l, c) for f in FAMILIES for l, c in (("fixed 10 LM iterations (the bench line)", 0),
                                                                    ("Ceres-default termination", 1))]:
    gmode, omode = MODES[fam]
    reg = 0.0 if fam == "nec" else 1e-13
    kw = dict(check_convergence=conv)
    if not conv:
        kw["max_num_iterations"] = 10
    opts = capi.default_options(**kw)
    oo = po.default_options(jacobian_mode=po.JAC_NUMERIC_CENTRAL, **kw)
    ang, it_equal, st_equal, n = [], 0, 0, 0
    for c, first in enumerate(range(0, B, CH)):
        m = min(CH, B - first)
        g = sim.generate(m, N, noise_type="anisotropic_inhomogeneous", noise_level=1.0, seed=1 + c, device=dev)
        cv = g.covs2.reshape(-1, 3, 3)
        with Batch.uniform(gmode, m, N) as b:
            if fam == "nec":
                b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3))
            elif fam == "sym":
                b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), cv, cv)
            else:
                b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), cv)
            res = b.solve(g.init_q, g.init_t, reg=reg, options=opts)
            gq, gi, gs = res.q.cpu().numpy(), res.iterations.cpu().numpy(), res.status.cpu().numpy()
        c9 = po.covs_to_colmajor9(cv.cpu().numpy())
        q, t, cost, it, st = po.solve_batch(
            omode, np.arange(m + 1, dtype=np.int64) * N, g.bvs1.reshape(-1, 3).cpu().numpy(),
            g.bvs2.reshape(-1, 3).cpu().numpy(), None if fam == "nec" else c9, c9 if fam == "sym" else None,
            reg, g.init_q.cpu().numpy(), g.init_t.cpu().numpy(), options=oo, num_threads=po.max_threads())
        ang.append(quat_angle(gq, q))
        it_equal += int((gi == it).sum())
        st_equal += int((gs == st).sum())
        n += m
        del g
    ang = np.concatenate(ang)
    print(json.dumps({"family": fam, "mode": label, "pairs": n, "corr": N, "against": "oracle, central-difference Jacobian (the reference's configuration)",
                      "max_rot_diff_rad": float(ang.max()), "p99_rot_diff_rad": float(np.percentile(ang, 99)),
                      "median_rot_diff_rad": float(np.median(ang)), "pairs_over_1e-6_rad": int((ang > 1e-6).sum()),
                      "iteration_counts_equal": it_equal, "termination_codes_equal": st_equal}), flush=True)
