"""One whole-chain PNEC::Solve per call from Python (pnec_amd.frame.FrameSolver.solve_raw, no marshalling): the C++ demo's
kind of pair (simulated, 512 correspondences) under schemes 2 and 0, then 300 frames of the bench's KITTI-like sequence
under scheme 2 -- to tell the Python loop's cost (~10 us) from the workload's (NOTES/round-6.md 6).  Run on the GPU box."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pnec_amd import capi, simulation as sim
from pnec_amd.frame import FrameSolver
import torch
g = sim.generate(1, 512, seed=5)
f1, f2 = g.bvs1[0].numpy().copy(), g.bvs2[0].numpy().copy()
c9 = np.ascontiguousarray(np.transpose(g.covs2[0].numpy(), (0, 2, 1)).reshape(-1, 9))
q0, t0 = g.init_q[0].numpy().copy(), g.init_t[0].numpy().copy()
oq, ot, mask = np.zeros(4), np.zeros(3), np.zeros(1024, dtype=np.uint8)   # (the KITTI-like frames below have up to ~700 correspondences)
for sch in (2, 0):
    o = capi.default_pipeline_options(eigensolver_scheme=sch)
    with FrameSolver(max_corr=1024) as fs:
        for _ in range(50): fs.solve_raw(512, f1, f2, c9, q0, t0, o, oq, ot, mask)
        ts = []
        for _ in range(300):
            t = time.perf_counter(); fs.solve_raw(512, f1, f2, c9, q0, t0, o, oq, ot, mask); ts.append(time.perf_counter() - t)
        print("scheme", sch, "python solve_raw median us", np.median(ts) * 1e6, "p10", np.percentile(ts, 10) * 1e6)
# the KITTI-like frames of the bench entry
offsets, F1, F2, C2, R_gt, t_gt, Q0, T0 = sim.generate_kitti_like(300, mean_corr=500, seed=3)
F1, F2, C2 = (x.numpy() for x in (F1, F2, C2))
C9 = np.ascontiguousarray(np.transpose(C2, (0, 2, 1)).reshape(-1, 9))
offsets = np.asarray(offsets)
o = capi.default_pipeline_options(eigensolver_scheme=2)
with FrameSolver(max_corr=1024) as fs:
    per = []
    for rep in range(2):
        per = []
        for pp in range(300):
            a, e = int(offsets[pp]), int(offsets[pp + 1])
            t = time.perf_counter(); fs.solve_raw(e - a, F1[a:e], F2[a:e], C9[a:e], Q0[pp].numpy(), T0[pp].numpy(), o, oq, ot, mask); per.append(time.perf_counter() - t)
    print("kitti-like frames, scheme 2: median us", np.median(per) * 1e6, "mean", np.mean(per) * 1e6, "p90", np.percentile(per, 90) * 1e6)
    # ... and with 10 % gross mismatches (what RANSAC is there for: most of a round's sixteen models get scored)
    rng = np.random.default_rng(0)
    bad = rng.random(len(F2)) < 0.10
    r = rng.standard_normal((len(F2), 3))
    F2m = F2.copy(); F2m[bad] = (r / np.linalg.norm(r, axis=1, keepdims=True))[bad]
    for rep in range(2):
        per = []
        for pp in range(300):
            a, e = int(offsets[pp]), int(offsets[pp + 1])
            t = time.perf_counter(); fs.solve_raw(e - a, F1[a:e], F2m[a:e], C9[a:e], Q0[pp].numpy(), T0[pp].numpy(), o, oq, ot, mask); per.append(time.perf_counter() - t)
    print("kitti-like frames with 10 % gross mismatches, scheme 2: median us", np.median(per) * 1e6, "mean", np.mean(per) * 1e6, "p90", np.percentile(per, 90) * 1e6)
