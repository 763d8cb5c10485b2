#!/usr/bin/env python3
"""Dump the inputs and both results (device, oracle twin) of the pairs whose weighted-stage rotations differ most
(the workload of tools/verify_frontend_literal.py) into gpurun_out/diag_worst.npz, for an off-line look at the
objective at both rotations.  Runs on the GPU box."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
B, N = 2048, 512
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
f1n, f2n, c2n = f1.cpu().numpy(), f2.cpu().numpy(), c2.cpu().numpy()
R_twin, t_twin = po.weighted_eigensolver_batch(offsets, f1n, f2n, c2n, Rn, tn, 1e-13, 10, device_early_exits=True)
Rd = np.stack([po.rot_from_quat(q) for q in qw])
e = np.array([np.radians(po.rotational_difference_deg(a, b)) for a, b in zip(Rd, R_twin)])
worst = np.argsort(-e)[:4]
print(worst, e[worst])
out = {}
for k, p in enumerate(worst):
    sl = slice(p * N, (p + 1) * N)
    out.update({f"f1_{k}": f1n[sl], f"f2_{k}": f2n[sl], f"c_{k}": c2n[sl], f"Rn_{k}": Rn[p], f"tn_{k}": tn[p], f"Rd_{k}": Rd[p], f"Ro_{k}": R_twin[p]})
np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "diag_worst.npz"), **out)
