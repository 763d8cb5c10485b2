#!/usr/bin/env python3
"""Throughput of the four residual families on the benchmark's batch shape:
B pairs x 512 correspondences x 10 LM iterations, plus the rotation difference against the oracle on a
sample.  Runs on the GPU box; prints one JSON object per family.  (HOST and SYM reuse the frame-2
covariances as frame-1 covariances: the arithmetic, not the data model, is what is timed.)
   python tools/bench_modes.py [B] [family: nec|target|host|sym|all] [cpl wpp ldsk]   (forced launch geometry: A/B runs)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
ONLY = sys.argv[2].lower() if len(sys.argv) > 2 else "all"
GEOM = [int(x) for x in sys.argv[3:6]] if len(sys.argv) > 5 else [0, 0, 0]
N = 512
dev = torch.device("cuda:0")
HBM_PEAK_GBS, FP64_VALU_PEAK_TFLOPS = 8000.0, 78.6          # as in bench.py
# algorithmic FP64 flop per correspondence: one full pass (eval_corr<MODE> + the 21 accumulating FMAs) | the cost-only
# pass that ends a solve at the iteration cap (eval_cost<MODE> + 1 FMA).  TARGET: counted from the ISA (62 FMA + 28 MUL +
# 1 rsq; bench.py); the others from the algebra of pnec_device.hpp with an FMA as 2 flop -- NEC drops the covariance
# products and the normalisation, HOST adds p = R f1, h = t x p, S h and the two extra Jacobian cross products, SYM both
# covariance terms.
FLOP = {"NEC": (112, 30), "PNEC target": (153, 67), "PNEC host": (206, 80), "PNEC symmetric": (227, 101)}
opts = capi.default_options(max_num_iterations=10, check_convergence=0, corr_per_lane=GEOM[0], waves_per_pair=GEOM[1],
                            lds_corr_per_lane=GEOM[2])
oo = po.default_options(jacobian_mode=po.JAC_NUMERIC_CENTRAL, max_num_iterations=10, check_convergence=0)


def quat_angle(a, b):
    d = np.clip(np.abs(np.sum(a * b, axis=-1)), 0, 1)
    v = np.linalg.norm(a[..., :3] * b[..., 3:4] - b[..., :3] * a[..., 3:4] - np.cross(a[..., :3], b[..., :3]), axis=-1)
    return 2 * np.arctan2(v, d)


for name, mode, omode in (("NEC", capi.MODE_NEC, po.MODE_NEC), ("PNEC target", capi.MODE_TARGET, po.MODE_TARGET),
                          ("PNEC host", capi.MODE_HOST, po.MODE_HOST), ("PNEC symmetric", capi.MODE_SYM, po.MODE_SYM)):
    if ONLY != "all" and ONLY not in name.lower():
        continue
    batch = Batch.uniform(mode, B, N)
    qs, ts, first = [], [], None
    for c0 in range(0, B, 10_000):
        m = min(10_000, B - c0)
        g = sim.generate(m, N, seed=1 + c0, device=dev)
        f1, f2, cv = g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3)
        if mode == capi.MODE_NEC:
            batch.fill(f1, f2, first_pair=c0, n_pairs=m)
        elif mode == capi.MODE_SYM:
            batch.fill(f1, f2, cv, cv, first_pair=c0, n_pairs=m)
        else:
            batch.fill(f1, f2, cv, first_pair=c0, n_pairs=m)
        qs.append(g.init_q); ts.append(g.init_t)
        if first is None:
            first = g
        else:
            del g
    q0, t0 = torch.cat(qs), torch.cat(ts)
    reg = 0.0 if mode == capi.MODE_NEC else 1e-13
    res = None
    for _ in range(10):
        res = batch.solve(q0, t0, reg=reg, options=opts, out=res)
    torch.cuda.synchronize()
    tt = []
    for _ in range(10):
        t = time.perf_counter()
        res = batch.solve(q0, t0, reg=reg, options=opts, out=res)
        torch.cuda.synchronize()
        tt.append(time.perf_counter() - t)
    el = float(np.median(tt))
    ns = 32
    c9 = po.covs_to_colmajor9(first.covs2[:ns].reshape(-1, 3, 3).cpu().numpy())
    oq = po.solve_batch(omode, np.arange(ns + 1, dtype=np.int64) * N, first.bvs1[:ns].reshape(-1, 3).cpu().numpy(),
                        first.bvs2[:ns].reshape(-1, 3).cpu().numpy(), None if mode == capi.MODE_NEC else c9,
                        c9 if mode == capi.MODE_SYM else None, reg, first.init_q[:ns].cpu().numpy(),
                        first.init_t[:ns].cpu().numpy(), options=oo)[0]
    ang = quat_angle(res.q[:ns].cpu().numpy(), oq)
    full, cost_only = FLOP[name]
    # the passes the kernel executed (cost-only after rejected steps and at the cap): one counted launch
    import ctypes as C
    cnt = np.zeros(16, dtype=np.uint64)
    flag = C.c_int32(0)
    capi.check(capi.lib().pnec_hip_work_counters(0, 1, cnt.ctypes.data, C.byref(flag)))
    o2 = capi.Options.from_buffer_copy(bytes(opts))
    o2.flags = 1
    batch.solve(q0, t0, reg=reg, options=o2)
    torch.cuda.synchronize()
    capi.check(capi.lib().pnec_hip_work_counters(0, 1, cnt.ctypes.data, C.byref(flag)))
    flops = float(cnt[13]) * full + float(cnt[14]) * cost_only
    gbs = batch.payload_bytes / el / 1e9
    print(json.dumps({"family": name, "pairs": B, "corr": N, "lm_iterations": 10, "ms": el * 1e3,
                      "solves_per_s": B / el, "launch": batch.describe_launch(opts),
                      "payload_bytes_per_pair": batch.payload_bytes // B,
                      "roofline": {"bound": "hbm", "bound_binding": "valu_fp64", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": batch.payload_bytes,
                                   "note": "read-once bytes over the wall time of one solve call (launch + kernel)",
                                   "valu": {"achieved": flops / el / 1e12, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                            "frac": flops / el / 1e12 / FP64_VALU_PEAK_TFLOPS,
                                            "flop_per_corr_full_pass": full, "flop_per_corr_cost_only_pass": cost_only,
                                            "passes_per_solve_full": float(cnt[13]) / (B * N), "passes_per_solve_cost_only": float(cnt[14]) / (B * N)}},
                      "max_rot_diff_vs_oracle_rad_32_pairs": float(ang.max())}), flush=True)
    batch.close()
    del batch, q0, t0, res
