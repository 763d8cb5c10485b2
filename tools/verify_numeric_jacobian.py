"""Config 4's in-bench parity sample (64 solves: one hypothesis of every pair, random t-hat starts), looked at four ways:

  device analytic   the production kernel (closed-form Jacobian)
  device numeric    PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL: the reference's differentiation on the device (verification)
  oracle numeric    the reference-faithful CPU path (central differences + Ceres LM policy)
  oracle numeric'   the same with the difference step scaled by 1 + 2^-10: the SAME derivative, other rounding of the quotient
  oracle analytic   the CPU twin of the production kernel

under (a) the throughput configuration -- exactly ten LM iterations, not converged from a start 180 degrees off -- and (b)
Ceres-default termination.  One JSON line per configuration: the distances (rad) between the five results per solve, so
that "the device is x rad from the reference path" can be read beside "the reference path is y rad from ITSELF under a
perturbation of its rounding".  Run on the GPU box: python tools/verify_numeric_jacobian.py [out.jsonl]
"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import pnec_oracle as po  # noqa: E402
from pnec_amd import Batch, capi  # noqa: E402
from pnec_amd import simulation as sim  # noqa: E402


def quat_angles(a, b):
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    d = np.abs(np.sum(a * b, axis=1)).clip(0, 1)
    v = np.stack([a[:, 3] * b[:, 0] - a[:, 0] * b[:, 3] - a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1],
                  a[:, 3] * b[:, 1] + a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] - a[:, 2] * b[:, 0],
                  a[:, 3] * b[:, 2] - a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] - a[:, 2] * b[:, 3]], 1)
    return 2.0 * np.arctan2(np.linalg.norm(v, axis=1), d)


def study(Bp=64, N=4096, H=64, seed=9, hyp_seed=5, device="cuda:0"):
    dev = torch.device(device)
    g = sim.generate(Bp, N, seed=seed, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(hyp_seed)
    hyp = torch.randn(Bp * H, 3, generator=gen, dtype=torch.float64, device=dev)
    hyp = hyp / hyp.norm(dim=1, keepdim=True)
    hyp[::H] = g.init_t
    picks = [(pp, (7 * pp + 3) % H) for pp in range(Bp)]
    idx = torch.tensor([pp * H + h for pp, h in picks], device=dev)
    hyp_s = hyp[idx].contiguous()                       # one start per pair: n_hyp = 1 with hyp_t [Bp, 3]
    f1, f2 = g.bvs1.cpu().numpy(), g.bvs2.cpu().numpy()
    c2 = g.covs2.cpu().numpy()
    q0, hs = g.init_q.cpu().numpy(), hyp_s.cpu().numpy()
    offsets = np.arange(Bp + 1, dtype=np.int64) * N
    c9 = po.covs_to_colmajor9(c2.reshape(-1, 3, 3))
    threads = po.usable_threads()
    out = []
    for name, kw in (("fixed_10_iterations", dict(max_num_iterations=10, check_convergence=0)),
                     ("ceres_default_termination", dict())):
        res = {}
        with Batch.uniform(capi.MODE_TARGET, Bp, N, device=dev.index) as b:
            b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
            for tag, flags in (("device_analytic", 0), ("device_numeric", capi.OPT_JACOBIAN_NUMERIC_CENTRAL)):
                r = b.solve(g.init_q, None, options=capi.default_options(flags=flags, **kw), hyp_t=hyp_s, n_hyp=1)
                torch.cuda.synchronize()
                res[tag] = (r.q.cpu().numpy(), r.iterations.cpu().numpy(), r.status.cpu().numpy())

        def cpu(jm, scale=1.0):
            po.set_numeric_step_scale(scale)
            try:
                o = po.solve_batch(po.MODE_TARGET, offsets, f1.reshape(-1, 3), f2.reshape(-1, 3), c9, None, 1e-13, q0, None,
                                   n_hyp=1, hyp_t=hs, options=po.default_options(jacobian_mode=jm, **kw), num_threads=threads)
            finally:
                po.set_numeric_step_scale(1.0)
            return o[0], o[3], o[4]
        res["oracle_numeric"] = cpu(po.JAC_NUMERIC_CENTRAL)
        res["oracle_numeric_step_x_1p001"] = cpu(po.JAC_NUMERIC_CENTRAL, 1.0 + 2.0 ** -10)
        res["oracle_analytic"] = cpu(po.JAC_ANALYTIC)
        SCALES = (1.0 + 2.0 ** -10, 1.0 - 2.0 ** -10, 1.0 + 2.0 ** -7, 2.0, 0.5)
        others = [cpu(po.JAC_NUMERIC_CENTRAL, sc)[0] for sc in SCALES]

        def d(a, b):
            return quat_angles(res[a][0], res[b][0])
        floor = np.max(np.stack([quat_angles(res["oracle_numeric"][0], oq) for oq in others]), axis=0)
        line = {"configuration": name, "n_solves": Bp, "pairs": f"{Bp} x {N}, hypothesis (7 p + 3) % {H} of pair p"}
        for a, b in (("device_analytic", "oracle_numeric"), ("device_numeric", "oracle_numeric"),
                     ("device_analytic", "oracle_analytic"), ("oracle_analytic", "oracle_numeric"),
                     ("oracle_numeric", "oracle_numeric_step_x_1p001"), ("device_analytic", "device_numeric")):
            x = d(a, b)
            line[f"{a}__vs__{b}"] = {"max_rad": float(x.max()), "median_rad": float(np.median(x)),
                                     "n_above_1e-6": int((x > 1e-6).sum()), "n_above_1e-7": int((x > 1e-7).sum()),
                                     "iteration_counts_equal": int((res[a][1] == res[b][1]).sum()),
                                     "termination_codes_equal": int((res[a][2] == res[b][2]).sum())}
        stable = floor <= 1e-7
        line["n_solves_whose_reference_path_is_stable_under_its_own_rounding_(<=1e-7)"] = int(stable.sum())
        for a in ("device_analytic", "device_numeric"):
            x = d(a, "oracle_numeric")
            line[f"{a}__vs__oracle_numeric__stable_solves_max_rad"] = float(x[stable].max()) if stable.any() else None
            line[f"{a}__vs__oracle_numeric__unstable_solves_max_rad"] = float(x[~stable].max()) if (~stable).any() else None
            # per unstable solve: is the device further from the reference path than the path is from itself?
            ratio = x[~stable] / np.maximum(floor[~stable], 1e-300)
            line[f"{a}__unstable_solves_distance_over_reference_self_distance_max"] = float(ratio.max()) if ratio.size else None
        line["reference_self_distance_unstable_solves_max_rad"] = float(floor[~stable].max()) if (~stable).any() else None
        line["lm_iterations_mean_device"] = float(res["device_analytic"][1].mean())
        line["step_scales_of_the_self_distance"] = list(SCALES)
        line["per_solve"] = {"reference_self_distance_rad": floor.tolist(),
                             "device_analytic_vs_oracle_numeric_rad": d("device_analytic", "oracle_numeric").tolist(),
                             "device_numeric_vs_oracle_numeric_rad": d("device_numeric", "oracle_numeric").tolist(),
                             "device_analytic_vs_oracle_analytic_rad": d("device_analytic", "oracle_analytic").tolist()}
        out.append(line)
    return out


if __name__ == "__main__":
    lines = study()
    for ln in lines:
        print(json.dumps(ln), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")
