#!/usr/bin/env python3
"""Several t-hat starts per pair on pairs that fit ONE wavefront (N <= 512): hypothesis-solves/s for H = 1, 2, 4, 16 at a
fixed number of solves (100 000), ten LM iterations.  With PNEC_SOLVE_GROUPS=0 every (pair, hypothesis) is a block of its
own (round 5); by default two hypotheses of a pair share a wavefront (lm_solve_pairhyp_kernel).
   python tools/bench_multihyp_small.py [corr] [solves]   (also for larger pairs: the several-wavefront group form)"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S = int(sys.argv[2]) if len(sys.argv) > 2 else max(4096, 100_000 * 512 // max(N, 512))
dev = torch.device("cuda:0")
out = {"corr": N, "solves": S, "PNEC_SOLVE_GROUPS": os.environ.get("PNEC_SOLVE_GROUPS", "1"), "rates_M_per_s": {}}
opts = capi.default_options(max_num_iterations=10, check_convergence=0)
for H in (1, 2, 4, 16):
    P = S // H
    g = sim.generate(P, N, seed=3, device=dev)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    hyp = g.init_t.repeat_interleave(H, dim=0) + 0.02 * torch.randn(P * H, 3, generator=gen, dtype=torch.float64, device=dev)
    hyp = hyp / hyp.norm(dim=1, keepdim=True)
    with Batch.uniform(capi.MODE_TARGET, P, N) as b:
        b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
        run = (lambda: b.solve(g.init_q, g.init_t, options=opts)) if H == 1 else (lambda: b.solve(g.init_q, None, options=opts, hyp_t=hyp, n_hyp=H))
        for _ in range(10): run()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): r = run()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 20 * 1e3
    out["rates_M_per_s"][str(H)] = P * H / ms / 1e3
    del g
print(json.dumps(out))
