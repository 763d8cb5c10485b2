#!/usr/bin/env python3
"""Reproduce the weighted stage of ONE pair of tools/bench_pipeline.py's workload with exactly the inputs it
has inside the batch (the device's RANSAC output and inlier batch), as a batch of one -- for a library built
with -DPNEC_FRONT_TRACE_ITER this prints the minimiser's iterations.   python tools/diag_weighted_pair.py B pair"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
B, PAIR = int(sys.argv[1]), int(sys.argv[2])
N = 512
dev = torch.device("cuda:0")
batch = Batch.uniform(capi.MODE_TARGET, B, N)
qs = []
keep = None
for c in range(0, B, 5000):
    m = min(5000, B - c)
    g = sim.generate(m, N, seed=1 + c, device=dev)
    bad = torch.rand(m, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c)) < 0.10
    rnd = torch.randn(m, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c + 1))
    rnd = rnd / rnd.norm(dim=-1, keepdim=True)
    g.bvs2 = torch.where(bad[..., None], rnd, g.bvs2)
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c, n_pairs=m)
    qs.append(g.init_q)
    if c <= PAIR < c + m:
        lp = PAIR - c
        keep = (g.bvs1[lp].cpu().numpy(), g.bvs2[lp].cpu().numpy(), g.covs2[lp].cpu().numpy())
q0 = torch.cat(qs)
qr, tr, mask, cnt, its = batch.ransac_eigensolver(q0, seed=1)
torch.cuda.synchronize()
msk = mask.reshape(B, N)[PAIR].cpu().numpy().astype(bool)
f1, f2, c2 = keep
f1i, f2i, c2i = f1[msk], f2[msk], c2[msk]
q1, t1 = qr[PAIR:PAIR + 1].cpu().numpy(), tr[PAIR:PAIR + 1].cpu().numpy()
print("inliers", int(msk.sum()), "ransac its", int(its[PAIR]))
sys.stdout.flush()
with Batch.uniform(capi.MODE_TARGET, 1, len(f1i)) as b:
    b.fill(f1i, f2i, c2i)
    qw, tw = b.weighted_eigensolver(q1, t1, 1e-13, 10)
Rr = po.rot_from_quat(q1[0])
w = np.array([po.weight(f1i[i], f2i[i], t1[0], Rr, c2i[i], 1e-13) * 1e-8 for i in range(len(f1i))])
print("weights: max %.3e median %.3e" % (w.max(), np.median(w)))
w2 = f2i * np.sqrt(w)[:, None]
r = po.eigensolver(f1i, w2, Rr)
print("oracle first minimisation iterations:", r[1:] if isinstance(r, tuple) else "")
Ro, to = po.weighted_eigensolver(f1i, f2i, c2i, Rr, t1[0])
print("device vs oracle weighted stage rot diff", np.radians(po.rotational_difference_deg(po.rot_from_quat(qw[0]), Ro)))
