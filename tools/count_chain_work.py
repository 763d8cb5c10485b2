#!/usr/bin/env python3
"""Algorithmic work of the chain's stages on the bench workloads, counted by a -DPNEC_WORK_COUNT build of the library
(tools/build_front_variant.sh count "-DPNEC_WORK_COUNT"; run this with PNEC_HIP_LIB pointing at it):

  PNEC_HIP_LIB=pnec_amd/csrc/build/var_count/libpnec_hip.so python tools/count_chain_work.py > profiles/chain_work_latest.json

Per workload (the synthetic kitti_all chain of bench.py --chain, and tools/bench_pipeline.py's 20 000 x 512 batch) and
per stage: quad-evaluations of the eigenvalue function, hypotheses prepared, tiles scored, correspondence terms of the
weighted stage's passes, ...  The counts are properties of the data (seeded) and of the algorithm, not of the timing;
bench.py combines them with the stage times it measures live (the key it checks: workload name, pairs, total
correspondences)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd import tracks as tk

NAMES = ["ransac_quad_evaluations", "ransac_hypotheses", "ransac_scored_tiles", "ransac_inlier_pass_corr",
         "es_on_inliers_quad_evaluations", "es_first_quad_evaluations", "weighted_inkernel_quad_evaluations",
         "weighted_table_corr", "weighted_cost_corr", "weighted_scf_corr", "weighted_bound_corr", "sums36_corr",
         "sums36_weighted_corr"]
L = capi.lib()


def counters(reset=True):
    out = np.zeros(16, dtype=np.uint64)
    flag = C.c_int32(0)
    capi.check(L.pnec_hip_work_counters(0, 1 if reset else 0, out.ctypes.data, C.byref(flag)))
    if not flag.value:
        sys.exit("this library was not built with -DPNEC_WORK_COUNT (set PNEC_HIP_LIB to the counting build)")
    return {n: int(out[i]) for i, n in enumerate(NAMES)}


def count(batch, q0, t0, scheme=0):
    batch.set_eigensolver_scheme(scheme)
    counters()
    qr, tr, mask, cnt, its = batch.ransac_eigensolver(q0, seed=1)
    torch.cuda.synchronize()
    r = counters()
    sel = batch.select(mask)
    qw, tw = sel.weighted_eigensolver(qr, tr, 1e-13, 10)
    torch.cuda.synchronize()
    w = counters()
    res = sel.solve(qw, tw)
    torch.cuda.synchronize()
    out = {"ransac_stage": {k: v for k, v in r.items() if v}, "weighted_stage": {k: v for k, v in w.items() if v},
           "ransac_iterations_sum": int(its.sum()), "inliers_sum": int(cnt.sum()),
           "refinement_lm_iterations_sum": int(res.iterations.sum())}
    sel.close()
    return out


dev = torch.device("cuda:0")
import bench  # noqa: E402  (the identity of the front-stage sources the counts belong to)
result = {"what": "algorithmic work per stage, counted by a -DPNEC_WORK_COUNT build (tools/count_chain_work.py)",
          "frontend_sources_sha256": bench.front_sources_sha256(), "workloads": {}}
# (1) bench.py --workload kitti_all --chain (synthetic stand-in, 10 % gross mismatches)
sizes = tk.kitti_all_sizes()
tr = tk.kitti_all_shard(0, len(sizes), device=dev, outlier_frac=0.10)
with Batch(capi.MODE_TARGET, tr.offsets) as b:
    b.fill(tr.bvs1, tr.bvs2, tr.covs)
    for sch in (0, 2):
        c = count(b, tr.init_q.contiguous(), tr.init_t.contiguous(), sch)
        c.update({"pairs": int(len(sizes)), "correspondences": int(sizes.sum()), "outliers": 0.10, "eigensolver_scheme": sch})
        result["workloads"]["kitti_all_chain" + ("" if sch == 0 else f"_scheme{sch}")] = c
del tr
# (2) tools/bench_pipeline.py 20000: 20 000 x 512, 10 % gross outliers
B, N = 20000, 512
batch = Batch.uniform(capi.MODE_TARGET, B, N)
qs, ts = [], []
for c0 in range(0, B, 5000):
    m = min(5000, B - c0)
    g = sim.generate(m, N, seed=1 + c0, device=dev)
    bad = torch.rand(m, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c0)) < 0.10
    rnd = torch.randn(m, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c0 + 1))
    rnd = rnd / rnd.norm(dim=-1, keepdim=True)
    g.bvs2 = torch.where(bad[..., None], rnd, g.bvs2)
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c0, n_pairs=m)
    qs.append(g.init_q); ts.append(g.init_t)
for sch in (0, 2):
    c = count(batch, torch.cat(qs), torch.cat(ts), sch)
    c.update({"pairs": B, "correspondences": B * N, "outliers": 0.10, "eigensolver_scheme": sch})
    result["workloads"]["sim20k_chain" + ("" if sch == 0 else f"_scheme{sch}")] = c
batch.close()
print(json.dumps(result, indent=1))
