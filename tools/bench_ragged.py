#!/usr/bin/env python3
"""Ragged-batch throughput (runs on the GPU box): B pairs whose correspondence counts are drawn
uniformly from [lo, hi], fixed 10 LM iterations, synthetic data as in bench.py.
   python tools/bench_ragged.py [B] [lo] [hi]
Prints one JSON line: solves/s, correspondences/s and the launch geometry histogram."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 700
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
counts = rng.integers(lo, hi + 1, size=B)
offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
# draw hi correspondences per pair in chunks, keep the first counts[i] of pair i
b = Batch(capi.MODE_TARGET, offsets)
qs, ts = [], []
chunk = 4000
for c0 in range(0, B, chunk):
    nb = min(chunk, B - c0)
    g = sim.generate(nb, hi, seed=11 + c0, device=dev)
    keep = torch.arange(hi, device=dev)[None, :] < torch.as_tensor(counts[c0:c0 + nb], device=dev)[:, None]
    b.fill(g.bvs1[keep], g.bvs2[keep], g.covs2[keep], first_pair=c0, n_pairs=nb)
    qs.append(g.init_q); ts.append(g.init_t)
    del g
q0, t0 = torch.cat(qs), torch.cat(ts)
opts = capi.default_options(max_num_iterations=10, check_convergence=0)
res = None
for _ in range(3):
    res = b.solve(q0, t0, options=opts, out=res)
torch.cuda.synchronize()
ts_ = []
for _ in range(7):
    t = time.perf_counter()
    res = b.solve(q0, t0, options=opts, out=res)
    torch.cuda.synchronize()
    ts_.append(time.perf_counter() - t)
t = float(np.median(ts_))
print(json.dumps({"workload": f"{B} ragged pairs, n ~ U[{lo},{hi}], 10 LM iterations", "ms": t * 1e3,
                  "solves_per_s": B / t, "corr_iterations_per_s": float(counts.sum()) * 10 / t,
                  "iterations": [int(res.iterations.min()), int(res.iterations.max())],
                  "cost_sum": float(res.cost.sum())}))
