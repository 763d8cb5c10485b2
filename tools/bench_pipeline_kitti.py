#!/usr/bin/env python3
"""The whole PNEC::Solve chain (one call) on the KITTI-like SYNTHETIC stream: all 23 190 ragged pairs of BASELINE
config 5 (265..700 correspondences, forward motion, 10 % gross outliers), one GPU.  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
P = int(sys.argv[1]) if len(sys.argv) > 1 else 23190
dev = torch.device("cuda:0")
offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=500, seed=11, device=dev)
M = f1.shape[0]
bad = torch.rand(M, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
rnd = torch.randn(M, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
rnd[:, 2] = rnd[:, 2].abs() + 1.0
f2 = torch.where(bad[:, None], rnd / rnd.norm(dim=-1, keepdim=True), f2)
off = offsets.cpu().numpy() if hasattr(offsets, "cpu") else np.asarray(offsets)
with Batch(capi.MODE_TARGET, off) as b:
    b.fill(f1, f2, c2)
    SCHEME = int(os.environ.get("PNEC_ES_SCHEME", "0"))   # include/pnec_hip.h pnec_hip_eigensolver_scheme
    O = capi.default_pipeline_options(eigensolver_scheme=SCHEME)
    def run():
        return b.solve_pipeline(q0, t0, O)
    run(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    ms = float(np.median(ts)) * 1e3
n = np.diff(off)
print(json.dumps({"eigensolver_scheme": SCHEME, "workload": "KITTI-like SYNTHETIC stream, %d ragged pairs (%d..%d correspondences, mean %.0f), 10 %% gross outliers, reference-default Options, one call" % (P, n.min(), n.max(), n.mean()),
                  "pairs_over_512": int((n > 512).sum()), "one_call_ms": ms, "pairs_per_s": P / ms * 1e3}))
