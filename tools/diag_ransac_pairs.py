#!/usr/bin/env python3
"""Dump the inputs and the device's RANSAC outputs of selected pairs of tools/verify_pipeline.py's workload
(same generator call, so P must be the P of the run in question).   python tools/diag_ransac_pairs.py P p1,p2,..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
P = int(sys.argv[1]); pairs = [int(x) for x in sys.argv[2].split(",")]
N = 512
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=1, device=dev)
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
with Batch.uniform(capi.MODE_TARGET, P, N) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    qr, tr, mask, cnt, its = b.ransac_eigensolver(g.init_q, seed=1)
    torch.cuda.synchronize()
out = {}
mask = mask.reshape(P, N)
for p in pairs:
    out.update({f"f1_{p}": g.bvs1[p].cpu().numpy(), f"f2_{p}": g.bvs2[p].cpu().numpy(), f"c2_{p}": g.covs2[p].cpu().numpy(),
                f"R0_{p}": g.init_R[p].cpu().numpy(), f"q0_{p}": g.init_q[p].cpu().numpy(), f"t0_{p}": g.init_t[p].cpu().numpy(),
                f"qr_{p}": qr[p].cpu().numpy(), f"tr_{p}": tr[p].cpu().numpy(), f"mask_{p}": mask[p].cpu().numpy(),
                f"cnt_{p}": cnt[p].cpu().numpy(), f"its_{p}": its[p].cpu().numpy()})
np.savez(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "diag_ransac_pairs.npz"), **out)
print("saved", pairs)
