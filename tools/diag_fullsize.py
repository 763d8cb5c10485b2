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
