#!/usr/bin/env python3
"""Does the one-call chain gain from running K slices of the batch on K streams (the tail of one slice's stage
filled by the other slices' work)?  Prints one JSON line per K.  usage: bench_pipeline_overlap.py [pairs] [corr]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda:0")


def make(first, m):
    b = Batch.uniform(capi.MODE_TARGET, m, N)
    qs, ts = [], []
    for c in range(first, first + m, 2500):   # the same global chunks whatever the slicing
        k = min(2500, first + m - c)
        g = sim.generate(k, N, seed=1 + c, device=dev)
        bad = torch.rand(k, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c)) < 0.10
        rnd = torch.randn(k, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c + 1))
        rnd = rnd / rnd.norm(dim=-1, keepdim=True)
        g.bvs2 = torch.where(bad[..., None], rnd, g.bvs2)
        b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c - first, n_pairs=k)
        qs.append(g.init_q); ts.append(g.init_t)
    return b, torch.cat(qs), torch.cat(ts)


ref = None
for K in (1, 2, 4, 8):
    m = B // K
    parts = [make(i * m, m) for i in range(K)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    opts = [capi.default_pipeline_options(first_pair_id=i * m) for i in range(K)]

    def run():
        outs = []
        for (b, q0, t0), s, o in zip(parts, streams, opts):
            with torch.cuda.stream(s):
                outs.append(b.solve_pipeline(q0, t0, o))
        return outs

    torch.cuda.synchronize(); run(); torch.cuda.synchronize()
    ts_ = []
    for _ in range(7):
        t = time.perf_counter(); outs = run(); torch.cuda.synchronize(); ts_.append(time.perf_counter() - t)
    q = torch.cat([o[0] for o in outs])
    if ref is None:
        ref = q
    print(json.dumps({"slices": K, "pairs": B, "ms": float(np.median(ts_)) * 1e3, "pairs_per_s": B / float(np.median(ts_)),
                      "bitwise_equals_one_slice": bool(torch.equal(q, ref))}), flush=True)
    del parts
