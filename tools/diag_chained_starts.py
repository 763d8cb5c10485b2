#!/usr/bin/env python3
"""PNEC_HIP_RANSAC_CHAINED_STARTS on the device against the checker's switch (runs on the GPU box): how many pairs agree
in mask and hypothesis count, and what the stage costs with the flag.   python tools/diag_chained_starts.py [pairs] [corr] [share]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
from oracle import pnec_oracle as oracle
P = int(sys.argv[1]) if len(sys.argv) > 1 else 600
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
OUT = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
dev = torch.device("cuda:0")
g = sim.generate(P, N, seed=3, device=dev)
bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < OUT
rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
f1, f2, R0 = g.bvs1.cpu().numpy(), g.bvs2.cpu().numpy(), g.init_R.cpu().numpy()
def qR(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
for s in (0, 2):
    with Batch.uniform(capi.MODE_TARGET, P, N) as b:
        b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
        b.set_eigensolver_scheme(s)
        rec = {}
        for flags in (0, capi.RANSAC_CHAINED_STARTS):
            b.set_ransac_flags(flags)
            out = b.ransac_eigensolver(g.init_q, seed=1); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = b.ransac_eigensolver(g.init_q, seed=1)
            torch.cuda.synchronize()
            rec[flags] = ((time.perf_counter() - t0) / 3 * 1e3, out)
    oracle.set_eigensolver_scheme(s)
    for flags in (0, 1):
        ms, (q, t, mask, cnt, its) = rec[flags]
        m_d, its_d, q_d = mask.cpu().numpy().reshape(P, N).astype(bool), its.cpu().numpy(), q.cpu().numpy()
        oracle.set_ransac_chained_starts(bool(flags))
        ok, ang, dits, jac = np.zeros(P, bool), np.zeros(P), np.zeros(P), np.zeros(P)
        for p in range(P):
            Ro, _, mo, io = oracle.ransac_eigensolver(f1[p], f2[p], R0[p], seed=1, pair_id=p)
            ok[p] = np.array_equal(mo, m_d[p]) and io == its_d[p]
            ang[p] = np.radians(oracle.rotational_difference_deg(Ro, qR(q_d[p])))
            dits[p] = its_d[p] - io
            jac[p] = (mo & m_d[p]).sum() / max(1, (mo | m_d[p]).sum())
        oracle.set_ransac_chained_starts(False)
        print(json.dumps({"scheme": s, "chained": flags, "stage_ms": ms, "mean_its_device": float(its_d.mean()), "identical": int(ok.sum()), "pairs": P,
                          "p99_rad_identical": float(np.percentile(ang[ok], 99)), "max_rad_others": float(ang[~ok].max()) if (~ok).any() else 0.0,
                          "min_mask_overlap_others": float(jac[~ok].min()) if (~ok).any() else 1.0,
                          "others_with_equal_counts": int((dits[~ok] == 0).sum())}))
oracle.set_eigensolver_scheme(0)
