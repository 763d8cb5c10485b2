"""Batch harness with the call pattern of the reference's `run_simulation` (src/run_simulation.cc:141-180,
348-391): for every experiment of a simulator folder run the ablation

    NEC            PNEC::Eigensolver (no RANSAC)                      pnec.cc:231-281
    NEC-LS         PNEC::NECCeresSolver from the initial pose          pnec.cc:394-411
    NEC PNEC-LS    PNEC::CeresSolver from the NEC result               pnec.cc:350-370
    PNEC only LS   PNEC::CeresSolver from the initial pose
    PNEC w/o LS    (the reference records the NEC result under this name, run_simulation.cc:171-174)
    PNEC           PNEC::CeresSolver from PNEC::WeightedEigensolver    pnec.cc:283-348

and write r_error.csv / t_error.csv / cost.csv in the reference's format.  All experiments of the
folder go through each stage in ONE device launch.  Input side as `ReadExperiments`
(src/simulation/sim_common.cc:109-236): relative pose = pose_1^-1 pose_2, bearings = normalised
points, frame-2 covariances through the unscented transform (K^-1 = I, kappa = 1), start pose = ground
truth perturbed by <= 0.01 rad / 0.01 (own RNG with a recorded seed: the C++ std::mt19937 +
uniform_real_distribution stream is not reproduced).

    python -m pnec_amd.run_simulation <experiment_folder> [--camera pinhole|omni] [--out DIR]
                                      [--init-scaling 1.0] [--seed 1]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

from . import capi
from .batch import Batch
from .frontend import CAMERA_OMNIDIRECTIONAL, CAMERA_PINHOLE, unscented_transform
from .io_formats import read_experiments, relative_poses, write_result_tables

METHODS = ("NEC", "NEC-LS", "NEC PNEC-LS", "PNEC only LS", "PNEC w/o LS", "PNEC")


def _quat_to_matrix(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - w * z); R[..., 0, 2] = 2 * (x * z + w * y)
    R[..., 1, 0] = 2 * (x * y + w * z); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - w * x)
    R[..., 2, 0] = 2 * (x * z - w * y); R[..., 2, 1] = 2 * (y * z + w * x); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _matrix_to_quat(R):
    from .simulation import matrix_to_quaternion_xyzw
    import torch
    return matrix_to_quaternion_xyzw(torch.as_tensor(np.asarray(R, dtype=np.float64))).numpy()


def rotational_difference_deg(R1, R2):
    """|log(R1' R2)| in degrees (common.cc:210-214)"""
    c = (np.einsum("eij,eij->e", R1, R2) - 1.0) * 0.5
    return np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))


def translational_difference_deg(t1, t2):
    """angle between directions, sign-agnostic (common.cc:216-235 with both_directions = true)"""
    n1, n2 = np.linalg.norm(t1, axis=1), np.linalg.norm(t2, axis=1)
    c = np.einsum("ei,ei->e", t1, t2) / np.where(n1 * n2 > 0, n1 * n2, 1.0)
    err = np.minimum(np.arccos(np.clip(c, -1, 1)), np.arccos(np.clip(-c, -1, 1)))
    return np.degrees(np.where(n1 < 1e-10, np.pi / 2, err))


def perturbed_start(R_gt, t_gt, rng, init_scaling=1.0):
    """sim_common.cc:205-231: rotation by sqrt(u) * 0.01 rad about a uniform axis, translation offset
    of length sqrt(u) * 0.01 in a uniform direction, applied on the left; translation normalised."""
    E = R_gt.shape[0]

    def sphere():
        theta = 2 * np.pi * rng.random(E)
        phi = np.arccos(1 - 2 * rng.random(E))
        return np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)], 1)

    axis = sphere()
    angle = np.sqrt(rng.random(E)) * 0.01 * init_scaling
    K = np.zeros((E, 3, 3))
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -axis[:, 2], axis[:, 1], axis[:, 2]
    K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -axis[:, 0], -axis[:, 1], axis[:, 0]
    s, c = np.sin(angle)[:, None, None], np.cos(angle)[:, None, None]
    dR = np.eye(3)[None] + s * K + (1 - c) * (K @ K)
    dt = sphere() * (np.sqrt(rng.random(E)) * 0.01 * init_scaling)[:, None]
    R0 = dR @ R_gt
    t0 = np.einsum("eij,ej->ei", dR, t_gt) + dt
    return R0, t0 / np.linalg.norm(t0, axis=1, keepdims=True)


def run(folder: str, camera: str = "pinhole", init_scaling: float = 1.0, seed: int = 1):
    ex = read_experiments(folder)
    E = len(ex["points_1"])
    counts = np.array([len(p) for p in ex["points_1"]], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)])
    R_gt, t_gt = relative_poses(ex["poses_1"], ex["poses_2"])
    p1 = np.concatenate(ex["points_1"])
    p2 = np.concatenate(ex["points_2"])
    c2 = np.concatenate(ex["covs_2"])
    b1 = p1 / np.linalg.norm(p1, axis=1, keepdims=True)
    cam = CAMERA_PINHOLE if camera == "pinhole" else CAMERA_OMNIDIRECTIONAL
    b2, cov = unscented_transform(p2, c2, None, 1.0, cam)      # device: UnscentedTransform + Unproject
    R0, t0 = perturbed_start(R_gt, t_gt, np.random.default_rng(seed), init_scaling)
    q0 = _matrix_to_quat(R0)

    sols = {}
    with Batch(capi.MODE_TARGET, offsets) as pb, Batch(capi.MODE_NEC, offsets) as nb:
        pb.fill(b1, b2, cov)
        nb.fill(b1, b2)
        q_nec, t_nec = nb.nec_eigensolver(q0)
        sols["NEC"] = (q_nec, t_nec)
        r = nb.solve(q0, t0, reg=0.0)
        sols["NEC-LS"] = (r.q, r.t)
        r = pb.solve(q_nec, t_nec, reg=1e-13)
        sols["NEC PNEC-LS"] = (r.q, r.t)
        r = pb.solve(q0, t0, reg=1e-13)
        sols["PNEC only LS"] = (r.q, r.t)
        q_it, t_it = pb.weighted_eigensolver(q_nec, t_nec, 1e-13, 10)
        sols["PNEC w/o LS"] = (q_nec, t_nec)   # sic: run_simulation.cc:171-174 stores nec_result here
        r = pb.solve(q_it, t_it, reg=1e-13)
        sols["PNEC"] = (r.q, r.t)
        results = {}
        for m in METHODS:
            q, t = sols[m]
            results[m] = {"r_error": rotational_difference_deg(R_gt, _quat_to_matrix(np.asarray(q))),
                          "t_error": translational_difference_deg(t_gt, np.asarray(t)),
                          "cost": pb.cost_function(np.asarray(q), np.asarray(t))}
    return results


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("folder")
    ap.add_argument("--camera", choices=("pinhole", "omni"), default="pinhole")
    ap.add_argument("--out", default=None, help="where to write r_error.csv / t_error.csv / cost.csv (default: the folder)")
    ap.add_argument("--init-scaling", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args(argv)
    res = run(args.folder, args.camera, args.init_scaling, args.seed)
    write_result_tables(args.out or args.folder, res)
    print(json.dumps({m: {"median_r_error_deg": float(np.median(res[m]["r_error"])),
                          "median_t_error_deg": float(np.median(res[m]["t_error"]))} for m in METHODS}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
