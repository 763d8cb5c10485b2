#!/usr/bin/env python3
"""Device weighted-eigensolver stage against the LITERAL oracle (pnec.cc:283-348 as it reads: the
eigensolver re-run in every round, 10 SCF steps every time) and against the oracle's early-exit twin,
over N pairs; prints one JSON object (-> profiles/r02_frontend_literal_parity.json).
Runs on the GPU box:  python tools/verify_frontend_literal.py [n_pairs] [n_corr]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
g = sim.generate(B, N, seed=4242, device=dev)
f1, f2, c2 = g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3)
offsets = np.arange(B + 1, dtype=np.int64) * N
with Batch(capi.MODE_TARGET, offsets) as b:
    b.fill(f1, f2, c2)
    qn, tn = b.nec_eigensolver(g.init_q)
    qw, tw = b.weighted_eigensolver(qn, tn, 1e-13, 10)
    torch.cuda.synchronize()
qn, tn, qw, tw = (x.cpu().numpy() for x in (qn, tn, qw, tw))
Rn = np.stack([po.rot_from_quat(q) for q in qn])
args = (offsets, f1.cpu().numpy(), f2.cpu().numpy(), c2.cpu().numpy(), Rn, tn, 1e-13, 10)
R_lit, t_lit = po.weighted_eigensolver_batch(*args, device_early_exits=False)
R_twin, t_twin = po.weighted_eigensolver_batch(*args, device_early_exits=True)


def rot_err(Ra, Rb):
    return np.array([np.radians(po.rotational_difference_deg(a, b)) for a, b in zip(Ra, Rb)])


Rd = np.stack([po.rot_from_quat(q) for q in qw])
e_lit, e_twin, e_ll = rot_err(Rd, R_lit), rot_err(Rd, R_twin), rot_err(R_twin, R_lit)
t_l = 1 - np.abs(np.sum(tw * t_lit, axis=1))
t_t = 1 - np.abs(np.sum(tw * t_twin, axis=1))
ang_t = np.arccos(np.clip(np.abs(np.sum(tw * t_lit, axis=1)), -1, 1))
stat = lambda e: {"max": float(e.max()), "p99": float(np.percentile(e, 99)), "median": float(np.median(e))}
print(json.dumps({
    "what": "weighted eigensolver + SCF stage (PNEC::WeightedEigensolver, 9 rounds), device vs oracle",
    "pairs": B, "correspondences": N, "cpu_threads": po.max_threads(),
    "device_vs_literal_oracle_rot_rad": stat(e_lit),
    "device_vs_early_exit_twin_rot_rad": stat(e_twin),
    "twin_vs_literal_rot_rad (the early exits alone)": stat(e_ll),
    "device_vs_literal_translation_angle_rad": stat(ang_t),
    "device_vs_literal_1_minus_abs_dot": stat(t_l), "device_vs_twin_1_minus_abs_dot": stat(t_t),
    "declared_bound_rad": 1e-7,
}), flush=True)
