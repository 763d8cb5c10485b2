#!/usr/bin/env python3
"""The RANSAC stage alone on tools/bench_pipeline.py's batch (20 000 x 512, 10 % gross mismatches by default): stage time
(median of 7, events) and a digest of everything it returns (poses, masks, counts, iteration counts).
   python tools/bench_ransac.py [pairs] [corr] [share of mismatches] [scheme]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
OUT = float(sys.argv[3]) if len(sys.argv) > 3 else 0.10
SCHEME = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
batch = Batch.uniform(capi.MODE_TARGET, B, N)
qs = []
for c in range(0, B, 5000):
    m = min(5000, B - c)
    g = sim.generate(m, N, seed=1 + c, device=dev)
    bad = torch.rand(m, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c)) < OUT
    rnd = torch.randn(m, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c + 1))
    g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c, n_pairs=m)
    qs.append(g.init_q)
q0 = torch.cat(qs)
batch.set_eigensolver_scheme(SCHEME)
out = batch.ransac_eigensolver(q0, seed=1); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = batch.ransac_eigensolver(q0, seed=1); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
h = hashlib.sha256()
for x in out:
    h.update(x.cpu().numpy().tobytes())
print(json.dumps({"pairs": B, "corr": N, "mismatches": OUT, "scheme": SCHEME, "launches": os.environ.get("PNEC_RANSAC_LAUNCHES", ""),
                  "ransac_stage_ms_median": float(np.median(ts)), "min": float(min(ts)), "mean_iterations": float(out[4].double().mean()),
                  "digest": h.hexdigest()[:16]}))
