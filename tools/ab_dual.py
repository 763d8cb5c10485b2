#!/usr/bin/env python3
"""A/B of the dual-step form of the refinement (PNEC_SOLVE_DUAL, read per process): results of this process' form for a
small batch, saved for comparison.   PNEC_SOLVE_DUAL=0|1 python tools/ab_dual.py out.npz [pairs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
out = sys.argv[1]; P = int(sys.argv[2]) if len(sys.argv) > 2 else 1001
g = sim.generate(P, 512, seed=1, device=torch.device("cuda:0"))
r = {}
with Batch.uniform(capi.MODE_TARGET, P, 512) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    for name, kw in (("fixed", dict(check_convergence=0, max_num_iterations=10)), ("ceres", dict())):
        res = b.solve(g.init_q, g.init_t, options=capi.default_options(**kw))
        torch.cuda.synchronize()
        for k in ("q", "t", "cost", "iterations", "status"):
            r[f"{name}_{k}"] = getattr(res, k).cpu().numpy()
np.savez(out, **r)
