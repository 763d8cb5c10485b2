#!/usr/bin/env python3
"""How much does the final pose depend on HOW the eigenvalue minimisation of the eigensolver stage iterates?

opengv is not in the reference tree, so the oracle and the device restate
opengv::relative_pose::eigensolver as "minimise lambda_min(M(R)) over the Cayley parameters" with a damped
Newton iteration converged to ~1e-12 rad.  opengv's own iteration, as far as its published source is remembered
[EXT, from memory, NOT verified against the source here]: steepest descent along the normalised gradient with an
adaptive step length lambda (start 0.01, doubled while it helps up to 0.08, halved while it does not), at most
50 iterations, stopped when lambda < 1e-5 -- i.e. it leaves the rotation within ~1e-5 of the minimiser.
This tool runs the reference's chain without RANSAC (eigensolver -> weighted eigensolver + SCF -> refinement)
on the CPU oracle twice per pair, once with the oracle's Newton eigensolver and once with that descent in its
place, and reports how far apart the rotations are after each stage: the refinement minimises the same
energy from either start, so what is left at the end is Ceres' stopping slack, not the eigensolver's.
CPU only (numpy + the oracle).   python tools/sensitivity_eigensolver_scheme.py [pairs] [correspondences]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pnec_oracle as po
from pnec_amd import simulation as sim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = sim.generate(P, N, seed=31)


def smallest_ev(f1, f2, v):
    M = po.compose_m(f1, f2, po.cayley_to_rot(v), skip_first=False)
    return float(np.linalg.eigvalsh(M)[0])


def descent_eigensolver(f1, f2, R0):
    """normalised steepest descent with an adaptive step (see the header: restated from memory)"""
    v = np.array(po.rot_to_cayley(R0), dtype=np.float64)
    lam, max_lam, mod, min_xtol = 0.01, 0.08, 2.0, 1e-5
    ev = smallest_ev(f1, f2, v)
    for it in range(50):
        h = 1e-7
        grad = np.array([(smallest_ev(f1, f2, v + h * e) - smallest_ev(f1, f2, v - h * e)) / (2 * h) for e in np.eye(3)])
        nrm = np.linalg.norm(grad)
        if not nrm > 0:
            break
        d = grad / nrm
        sp = v - lam * d
        sev = smallest_ev(f1, f2, sp)
        if it == 0:
            while sev < ev:
                ev = sev
                if lam * mod > max_lam:
                    break
                lam *= mod
                sp = v - lam * d
                sev = smallest_ev(f1, f2, sp)
        while sev > ev and lam > 1e-12:
            lam /= mod
            sp = v - lam * d
            sev = smallest_ev(f1, f2, sp)
        v, ev = sp, sev
        if lam < min_xtol:
            break
    return po.cayley_to_rot(v)


def chain(f1, f2, c2, R_es):
    M = po.compose_m(f1, f2, R_es, skip_first=True)
    t_es = po.translation_from_m(M)
    Rw, tw = po.weighted_eigensolver(f1, f2, c2, R_es, t_es)
    s = po.solve(po.MODE_TARGET, f1, f2, c2, None, 1e-13, po.quat_from_rot(Rw), tw, po.default_options())
    return Rw, s.R


rad = lambda A, B: float(np.radians(po.rotational_difference_deg(A, B)))
d_es, d_w, d_ls = [], [], []
for p in range(P):
    f1, f2, c2, R0 = g.bvs1[p].numpy(), g.bvs2[p].numpy(), g.covs2[p].numpy(), g.init_R[p].numpy()
    r = po.eigensolver(f1, f2, R0)
    Rn = r[0] if isinstance(r, tuple) else r
    Rd = descent_eigensolver(f1, f2, R0)
    Rw_n, Rl_n = chain(f1, f2, c2, Rn)
    Rw_d, Rl_d = chain(f1, f2, c2, Rd)
    d_es.append(rad(Rn, Rd)); d_w.append(rad(Rw_n, Rw_d)); d_ls.append(rad(Rl_n, Rl_d))
st = lambda x: {"max": float(np.max(x)), "p99": float(np.percentile(x, 99)), "median": float(np.median(x))}
print(json.dumps({"what": "rotation difference between the chain run with the oracle's Newton eigensolver and with an opengv-style normalised "
                          "steepest descent (restated from memory, [EXT] unverified) in its place, CPU oracle on both sides, no RANSAC",
                  "pairs": P, "correspondences": N,
                  "after_eigensolver_rad": st(d_es), "after_weighted_eigensolver_rad": st(d_w), "after_refinement_rad": st(d_ls)}))
