#!/usr/bin/env python3
"""Per-hypothesis comparison of the RANSAC eigensolver's sample models, device minimiser vs oracle minimiser, for
the pairs dumped by tools/diag_ransac_pairs.py: finds the hypotheses whose two minimisers end at different
rotations.  Runs on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "diag_ransac_pairs.npz"))
pairs = sorted({int(k.split("_")[1]) for k in d.files if k.startswith("f1_")})
H, SS = 64, 10
L = po.lib()
for p in pairs:
    f1, f2, R0 = d[f"f1_{p}"], d[f"f2_{p}"], d[f"R0_{p}"]
    n = len(f1)
    v0 = po.rot_to_cayley(R0)
    sels, starts = [], []
    for h in range(H):
        sel, draw = [], 0
        while len(sel) < SS:
            idx = int(L.pnec_oracle_rng_uniform(1, p, h, draw) * n); draw += 1
            idx = min(idx, n - 1)
            if idx not in sel: sel.append(idx)
        v = np.array([v0[c] + (L.pnec_oracle_rng_uniform(1, p, h, 1000 + c) - 0.5) * 2.0 * 0.01 for c in range(3)])
        sels.append(sel); starts.append(po.cayley_to_rot(v))
    s1 = np.concatenate([f1[s] for s in sels]); s2 = np.concatenate([f2[s] for s in sels])
    q0 = np.stack([po.quat_from_rot(R) for R in starts])
    with Batch.uniform(capi.MODE_NEC, H, SS) as b:
        b.fill(s1, s2)
        qd, td = b.nec_eigensolver(q0)
    rows = []
    for h in range(H):
        Ro = po.eigensolver(f1[sels[h]], f2[sels[h]], starts[h])
        Ro = Ro[0] if isinstance(Ro, tuple) else Ro
        Rd = po.rot_from_quat(qd[h])
        diff = np.radians(po.rotational_difference_deg(Ro, Rd))
        if diff > 1e-8:
            # objective lambda_min(M(R)) of the sample at both
            def lam(R):
                M = po.compose_m(f1[sels[h]], f2[sels[h]], R, skip_first=False)
                return float(np.linalg.eigvalsh(M)[0])
            rows.append({"hyp": h, "rot_diff_rad": float(diff), "lambda_min_oracle": lam(Ro), "lambda_min_device": lam(Rd),
                         "start_to_oracle_rad": float(np.radians(po.rotational_difference_deg(starts[h], Ro))),
                         "start_to_device_rad": float(np.radians(po.rotational_difference_deg(starts[h], Rd)))})
    print(p, "hypotheses with different minimisers:", rows)
