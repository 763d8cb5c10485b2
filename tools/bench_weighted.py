#!/usr/bin/env python3
"""The weighted stage alone on tools/bench_pipeline.py's batch (20 000 x 512, 10 % gross mismatches, after RANSAC +
InlierExtraction): stage time (median of 7, events) and, with PNEC_HIP_TRACE_FRONT=1, the kernel's phase clocks on
stderr.   python tools/bench_weighted.py [pairs] [corr]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")
batch = Batch.uniform(capi.MODE_TARGET, B, N)
qs = []
for c in range(0, B, 5000):
    m = min(5000, B - c)
    g = sim.generate(m, N, seed=1 + c, device=dev)
    bad = torch.rand(m, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c)) < 0.10
    rnd = torch.randn(m, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c + 1))
    g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c, n_pairs=m)
    qs.append(g.init_q)
q0 = torch.cat(qs)
tr = os.environ.pop("PNEC_HIP_TRACE_FRONT", None)
qr, trr, mask, cnt, its = batch.ransac_eigensolver(q0, seed=1)
sel = batch.select(mask)
sel.weighted_eigensolver(qr, trr, 1e-13, 10); torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = sel.weighted_eigensolver(qr, trr, 1e-13, 10); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
import hashlib
dig = hashlib.sha256(out[0].cpu().numpy().tobytes() + out[1].cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps({"pairs": B, "corr": N, "weighted_stage_ms_median": float(np.median(ts)), "min": float(min(ts)), "digest": dig}))
if tr:
    os.environ["PNEC_HIP_TRACE_FRONT"] = tr
    sel.weighted_eigensolver(qr, trr, 1e-13, 10); torch.cuda.synchronize()
