#!/usr/bin/env python3
"""Digest (sha256) of the refinement's results over the benchmark's batch -- quaternions, translations, costs, iteration
counts, termination codes -- in the fixed-10-iterations mode of the bench line and with Ceres-default termination, for
A/B builds that must not move a bit (run under PNEC_HIP_LIB=<variant> and compare).  Runs on the GPU box.
   python tools/digest_solve.py [pairs] [max_num_iterations, ...]"""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
caps = [int(x) for x in sys.argv[2:]] or [10]
N, CH = 512, 10000
dev = torch.device("cuda:0")
out = {"pairs": B, "lib": os.environ.get("PNEC_HIP_LIB", "default")}
modes = [("fixed%d" % c, dict(check_convergence=0, max_num_iterations=c)) for c in caps] + [("ceres_default", dict(check_convergence=1))]
for label, kw in modes:
    h = hashlib.sha256()
    ms = 0.0
    for c, first in enumerate(range(0, B, CH)):
        m = min(CH, B - first)
        g = sim.generate(m, N, noise_type="anisotropic_inhomogeneous", noise_level=1.0, seed=1 + c, device=dev)
        with Batch.uniform(capi.MODE_TARGET, m, N) as b:
            b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
            opts = capi.default_options(**kw)
            res = b.solve(g.init_q, g.init_t, reg=1e-13, options=opts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                res = b.solve(g.init_q, g.init_t, reg=1e-13, options=opts)
            torch.cuda.synchronize()
            ms += (time.perf_counter() - t0) / 5 * 1e3
            for x in (res.q, res.t, res.cost, res.iterations, res.status):
                h.update(x.cpu().numpy().tobytes())
    out[label] = {"digest": h.hexdigest()[:16], "ms": round(ms, 4)}
print(json.dumps(out))
