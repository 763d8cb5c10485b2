#!/usr/bin/env python3
"""Scratch diagnostics of the full-size batch (round 1): per-pair iteration counts and termination codes of the
bench workload.  Runs on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
B, N, chunk = 100_000, 512, 10_000
batch = Batch.uniform(capi.MODE_TARGET, B, N)
qs, ts = [], []
for c in range(B // chunk):
    g = sim.generate(chunk, N, seed=1000 + c, device="cuda:0")
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c * chunk, n_pairs=chunk)
    qs.append(g.init_q); ts.append(g.init_t); del g
q0, t0 = torch.cat(qs), torch.cat(ts)
res = batch.solve(q0, t0)
again = batch.solve(res.q, res.t)
torch.cuda.synchronize()
print("status hist", torch.bincount(res.status, minlength=7).tolist())
print("iters hist", torch.bincount(res.iterations).tolist())
print("again status hist", torch.bincount(again.status, minlength=7).tolist())
print("again iters hist", torch.bincount(again.iterations).tolist())
dq = (res.q * again.q).sum(-1).abs().clamp(max=1.0)
ang = 2 * torch.acos(dq)
print("idempotence max ang", float(ang.max()), "n>1e-6", int((ang > 1e-6).sum()))
bad = torch.nonzero(res.status > 2).flatten()[:10]
print("bad idx", bad.tolist(), res.status[bad].tolist(), res.iterations[bad].tolist(), res.cost[bad].tolist())

# dump a few non-converged pairs + compare with the CPU oracle on this box
from oracle import pnec_oracle as po
bad = torch.nonzero(res.status > 2).flatten()[:6].tolist()
dump = {}
g_cache = {}
for p in bad:
    c, j = divmod(p, chunk)
    if c not in g_cache:
        g_cache[c] = sim.generate(chunk, N, seed=1000 + c, device="cuda:0")
    g = g_cache[c]
    f1, f2, c2 = g.bvs1[j].cpu().numpy(), g.bvs2[j].cpu().numpy(), g.covs2[j].cpu().numpy()
    qi, ti = g.init_q[j].cpu().numpy(), g.init_t[j].cpu().numpy()
    for jm in (0, 1):
        s = po.solve(po.MODE_TARGET, f1, f2, c2, None, 1e-13, qi, ti, po.default_options(jacobian_mode=jm))
        gq = res.q[p].cpu().numpy()
        print("pair", p, "jac", jm, "oracle iters", s.iterations, po.TERM_NAMES[s.status], "gpu iters", int(res.iterations[p]),
              "rot diff rad", np.radians(po.rotational_difference_deg(s.R, po.rot_from_quat(gq))), "cost", s.cost, float(res.cost[p]))
    dump[f"p{p}_f1"], dump[f"p{p}_f2"], dump[f"p{p}_c2"], dump[f"p{p}_q0"], dump[f"p{p}_t0"] = f1, f2, c2, qi, ti
    dump[f"p{p}_Rgt"], dump[f"p{p}_tgt"] = g.R_gt[j].cpu().numpy(), g.t_gt[j].cpu().numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bad_pairs.npz"), **dump)
