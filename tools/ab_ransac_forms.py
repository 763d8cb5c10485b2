#!/usr/bin/env python3
"""RANSAC-stage forms against each other on one box: the whole chain over a ragged batch (the shapes of
tests/test_chain_scale_gpu.py's bitwise test) or a uniform one, with the form forced through PNEC_RANSAC_FORM
(1 one pair per wavefront, 2 two pairs -- the split form, 3, was removed in round 5; read once per process, so one
process per form).  Prints a digest of (q, t, mask, count) and the time per call.
   PNEC_RANSAC_FORM=3 python tools/ab_ransac_forms.py ragged 6001 | uniform 20000 [outlier_fraction]"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

kind = sys.argv[1] if len(sys.argv) > 1 else "ragged"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 6001
frac = float(sys.argv[3]) if len(sys.argv) > 3 else (0.15 if kind == "ragged" else 0.10)
dev = torch.device("cuda:0")
if kind == "ragged":
    rng = np.random.default_rng(77)
    counts = rng.integers(60, 640, size=P).astype(np.int64)
    counts[:8] = [5, 9, 10, 11, 64, 512, 513, 639]
else:
    counts = np.full(P, 512, dtype=np.int64)
off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
nmax = int(counts.max())
parts = []
for c0 in range(0, P, 1000):
    m = min(1000, P - c0)
    g = sim.generate(m, nmax, seed=900 + c0, device=dev)
    keep = torch.arange(nmax, device=dev)[None, :] < torch.as_tensor(counts[c0:c0 + m], device=dev)[:, None]
    bad = torch.rand(m, nmax, device=dev, generator=torch.Generator(device=dev).manual_seed(c0)) < frac
    rnd = torch.randn(m, nmax, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c0 + 1))
    b2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
    parts.append((g.bvs1[keep], b2[keep], g.covs2[keep], g.init_q, g.init_t))
f1, f2, cv, q0, t0 = (torch.cat([p[i] for p in parts]) for i in range(5))
with Batch(capi.MODE_TARGET, off) as b:
    b.fill(f1, f2, cv)
    q, t, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
    torch.cuda.synchronize()
    reps = 5
    t_0 = time.perf_counter()
    for _ in range(reps):
        b.solve_pipeline(q0, t0, want_inliers=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t_0) / reps * 1e3
    qr, tr, mk, cn, its = b.ransac_eigensolver(q0, seed=1)
    torch.cuda.synchronize()
    t_0 = time.perf_counter()
    for _ in range(reps):
        b.ransac_eigensolver(q0, seed=1)
    torch.cuda.synchronize()
    ms_r = (time.perf_counter() - t_0) / reps * 1e3
h = hashlib.sha256()
for x in (q, t, mask, cnt, qr, tr, mk, cn, its):
    h.update(x.cpu().numpy().tobytes())
if os.environ.get("AB_DUMP"):
    np.savez(os.environ["AB_DUMP"], q=q.cpu().numpy(), t=t.cpu().numpy(), mask=mask.cpu().numpy(), cnt=cnt.cpu().numpy(),
             qr=qr.cpu().numpy(), tr=tr.cpu().numpy(), mk=mk.cpu().numpy(), cn=cn.cpu().numpy(), its=its.cpu().numpy(), off=off)
print(json.dumps({"form": os.environ.get("PNEC_RANSAC_FORM", "auto"), "kind": kind, "pairs": P, "outliers": frac,
                  "chain_ms": ms, "ransac_stage_ms": ms_r, "mean_inliers": float(cnt.double().mean()),
                  "mean_ransac_iterations": float(its.double().mean()), "max_ransac_iterations": int(its.max()),
                  "digest": h.hexdigest()}))
