"""Externally supplied tracks: the on-disk input format of the batch solver, its loader, and the
synthetic stand-in for BASELINE config 5 ("all KITTI 00-10 frame pairs") when no tracks exist.

FORMAT (``.npz``, numpy, little-endian; everything float64 unless noted) -- one file holds a run of
consecutive frame pairs, ragged sizes:

  ==============  =================  ==========================================================
  key             shape              meaning (reference convention)
  ==============  =================  ==========================================================
  ``offsets``     int64 [P+1]        pair p owns correspondences [offsets[p], offsets[p+1])
  ``bvs1``        [M,3]              unit bearing vectors, frame 1 ("host")  (opengv::bearingVectors_t)
  ``bvs2``        [M,3]              unit bearing vectors, frame 2 ("target")
  ``covs``        [M,3,3]            bearing covariances of frame 2 (``projected_covs`` of
                                     PNEC::Solve, src/rel_pose_estimation/pnec.cc:69-75); symmetric
  ``init_q``      [P,4] xyzw         start orientation per pair (Eigen coeffs() order)
  ``init_t``      [P,3]              start translation per pair (any non-zero vector)
  ``covs_host``   [M,3,3] optional   frame-1 covariances (symmetric residual only)
  ``sequence``    int32 [P] optional KITTI sequence id of each pair (bookkeeping)
  ==============  =================  ==========================================================

This is what the reference's VO front end hands to ``PNEC::Solve`` per frame
(``Frame2Frame::PNECAlign``, src/rel_pose_estimation/frame2frame.cc:122-141: bearing vectors from
``KeyPoint::Unproject``, covariances from ``UnscentedTransform``, the previous relative pose as the
start), concatenated over the frames of a sequence.  ``save_tracks`` / ``load_tracks`` round-trip it;
``load_tracks(..., first_pair, last_pair)`` returns one rank's contiguous shard.

WHAT A KITTI (KLT) EXPORT MUST FOLLOW.  In the reference's odometry a keypoint's image covariance is the top-left 2x2
block of the INVERSE of the SE(2) Gauss-Newton Hessian of its tracked patch (H_se2 = J_se2' J_se2 over the pattern's
pixels, columns = d/dx, d/dy, d/dangle; include/features/tracking/pnec_patch.h:128-136: ``Cov << H_se2_inv(0,0) ...``)
DIVIDED BY uncertainty_scaling = 10 (klt_patch_optical_flow.h:377,381,458), taken at the finest pyramid level of the
FRAME-2 patch; it is unitless up to the image noise variance the Hessian leaves out, which is harmless -- the PNEC
weights are invariant to one global scale of the covariances (up to the regularisation 1e-13).  ``KeyPoint::Unproject``
(src/frames/keypoints.cc:49-62) then makes the bearing = normalised K^-1 (u, v, 1) and its 3x3 covariance =
``UnscentedTransform(mu, Sigma_2x2, K^-1, kappa = 1, Pinhole)`` (src/common/common.cc:467-525).  An export writes, per
consecutive frame pair, the matched tracks' bearings of both frames (the SAME track id at the same row of ``bvs1`` and
``bvs2``; optional ``ids1`` / ``ids2`` int64 [M] let ``check`` verify that), the frame-2 bearing covariances, and the
previous relative pose as the start (``Frame2Frame``: constant-velocity prior).  ``python -m pnec_amd.tracks check f.npz``
validates a file; ``python -m pnec_amd.tracks from-experiments <folder> out.npz`` converts the one on-disk product of the
reference that exists here, the simulator's CSV folder (src/simulation/experiments.cc:131-172).

KITTI itself is not in this environment: ``kitti_all_sizes`` / ``kitti_all_shard`` build a SYNTHETIC
stand-in with the real odometry sequences' lengths (pairs = frames - 1 per sequence, 23 190 in
total) and KITTI-like geometry (``simulation.generate_kitti_like``).  Pair p's data is a function of
(seed, p) only -- chunks of ``CHUNK`` pairs with their own seeds -- so any rank can build any range
and the whole set is identical for every world size.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# frames per KITTI odometry sequence 00..10 (the public dataset's sizes); pairs = frames - 1
KITTI_FRAMES = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101, 4071, 1591, 1201)
CHUNK = 512  # pairs per generation chunk (seeded independently)


@dataclass
class Tracks:
    offsets: np.ndarray          # int64 [P+1], offsets[0] == 0
    bvs1: object                 # [M,3]   numpy or torch
    bvs2: object                 # [M,3]
    covs: object                 # [M,3,3]
    init_q: object               # [P,4]
    init_t: object               # [P,3]
    covs_host: object = None     # [M,3,3] or None
    sequence: np.ndarray | None = None
    ids1: np.ndarray | None = None   # int64 [M] optional: track id of each row in frame 1 / frame 2 (must be equal)
    ids2: np.ndarray | None = None
    data: str = "tracks"         # provenance label for reports ("synthetic ..." or the file name)

    @property
    def n_pairs(self) -> int:
        return len(self.offsets) - 1

    @property
    def sizes(self) -> np.ndarray:
        return np.diff(self.offsets)


def save_tracks(path: str, tr: Tracks) -> None:
    def host(a):
        return None if a is None else (a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a))
    arrays = {"offsets": np.asarray(tr.offsets, dtype=np.int64), "bvs1": host(tr.bvs1), "bvs2": host(tr.bvs2),
              "covs": host(tr.covs), "init_q": host(tr.init_q), "init_t": host(tr.init_t)}
    if tr.covs_host is not None:
        arrays["covs_host"] = host(tr.covs_host)
    if tr.sequence is not None:
        arrays["sequence"] = np.asarray(tr.sequence, dtype=np.int32)
    if tr.ids1 is not None and tr.ids2 is not None:
        arrays["ids1"], arrays["ids2"] = np.asarray(tr.ids1, dtype=np.int64), np.asarray(tr.ids2, dtype=np.int64)
    np.savez(path, **arrays)


def validate(tr: Tracks) -> None:
    """Shape / consistency checks of a Tracks object; raises ValueError with the offending key."""
    off = np.asarray(tr.offsets)
    if off.ndim != 1 or len(off) < 1 or off[0] != 0 or (np.diff(off) < 0).any():
        raise ValueError("offsets must be int64 [P+1], non-decreasing, offsets[0] == 0")
    P, M = len(off) - 1, int(off[-1])
    want = {"bvs1": (M, 3), "bvs2": (M, 3), "covs": (M, 3, 3), "init_q": (P, 4), "init_t": (P, 3)}
    if tr.covs_host is not None:
        want["covs_host"] = (M, 3, 3)
    for k, shape in want.items():
        got = tuple(getattr(tr, k).shape)
        if got != shape:
            raise ValueError(f"{k}: expected shape {shape}, got {got}")
    if tr.sequence is not None and len(tr.sequence) != P:
        raise ValueError("sequence must have one entry per pair")


def check(tr: Tracks, min_corr: int = 10) -> dict:
    """Content checks of a tracks object beyond its shapes (``validate``): unit-norm bearings, finite values, symmetric
    positive semi-definite covariances, usable start poses, matched ids, per-pair counts.  -> a report dict whose
    ``ok`` says whether the solver can be fed with it; ``problems`` lists what is wrong (nothing is raised)."""
    validate(tr)
    host = lambda a: a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b1, b2, cv, q0, t0 = (host(x) for x in (tr.bvs1, tr.bvs2, tr.covs, tr.init_q, tr.init_t))
    sizes = np.diff(np.asarray(tr.offsets))
    problems = []
    rep = {"pairs": int(len(sizes)), "correspondences": int(sizes.sum()),
           "corr_per_pair_min_mean_max": [int(sizes.min()) if len(sizes) else 0, float(sizes.mean()) if len(sizes) else 0.0,
                                          int(sizes.max()) if len(sizes) else 0]}
    for name, b in (("bvs1", b1), ("bvs2", b2)):
        if not np.isfinite(b).all():
            problems.append(f"{name}: non-finite values")
        dev = np.abs(np.linalg.norm(b, axis=1) - 1.0)
        rep[f"{name}_max_norm_deviation"] = float(dev.max()) if len(dev) else 0.0
        if len(dev) and dev.max() > 1e-9:
            problems.append(f"{name}: {int((dev > 1e-9).sum())} bearing vectors are not unit length (max deviation {dev.max():.3e})")
    covs = [("covs", cv)] + ([("covs_host", host(tr.covs_host))] if tr.covs_host is not None else [])
    for name, c in covs:
        if not np.isfinite(c).all():
            problems.append(f"{name}: non-finite values")
            continue
        if len(c):
            scale = np.abs(c).max(axis=(1, 2)) + 1e-300
            asym = np.abs(c - c.transpose(0, 2, 1)).max(axis=(1, 2)) / scale
            w = np.linalg.eigvalsh(0.5 * (c + c.transpose(0, 2, 1)))
            rep[f"{name}_max_relative_asymmetry"] = float(asym.max())
            rep[f"{name}_min_eigenvalue_over_max"] = float((w[:, 0] / (w[:, 2] + 1e-300)).min())
            rep[f"{name}_median_std"] = float(np.median(np.sqrt(np.maximum(w[:, 2], 0.0))))
            if asym.max() > 1e-9:
                problems.append(f"{name}: {int((asym > 1e-9).sum())} matrices are not symmetric")
            if (w[:, 0] < -1e-9 * w[:, 2]).any():
                problems.append(f"{name}: {int((w[:, 0] < -1e-9 * w[:, 2]).sum())} matrices are not positive semi-definite")
            if (w[:, 2] <= 0).any():
                problems.append(f"{name}: {int((w[:, 2] <= 0).sum())} matrices are zero")
    if len(q0) and (not np.isfinite(q0).all() or (np.linalg.norm(q0, axis=1) < 1e-12).any()):
        problems.append("init_q: zero or non-finite quaternions")
    if len(t0) and (not np.isfinite(t0).all() or (np.linalg.norm(t0, axis=1) < 1e-12).any()):
        problems.append("init_t: zero or non-finite start translations")
    if (tr.ids1 is None) != (tr.ids2 is None):
        problems.append("ids1 / ids2: only one of the two is present")
    if tr.ids1 is not None and tr.ids2 is not None:
        i1, i2 = np.asarray(tr.ids1), np.asarray(tr.ids2)
        if i1.shape != (int(sizes.sum()),) or i2.shape != i1.shape:
            problems.append("ids1 / ids2: expected int64 [M]")
        else:
            rep["matched_ids_equal"] = bool((i1 == i2).all())
            if not rep["matched_ids_equal"]:
                problems.append(f"ids: {int((i1 != i2).sum())} rows pair different track ids in the two frames")
            off = np.asarray(tr.offsets)
            dup = sum(len(np.unique(i1[off[p]:off[p + 1]])) != off[p + 1] - off[p] for p in range(len(sizes)))
            if dup:
                problems.append(f"ids: {dup} pairs list a track id more than once")
    small = int((sizes < min_corr).sum())
    rep["pairs_below_min_corr"] = small
    if small:
        problems.append(f"{small} pairs have fewer than {min_corr} correspondences (the reference skips a frame below "
                        "Options::min_matches_ = 30; RANSAC needs ransac_sample_size_ = 10)")
    rep["problems"], rep["ok"] = problems, not problems
    return rep


def from_experiments(folder: str, camera: str = "pinhole", init_scaling: float = 1.0, seed: int = 1) -> Tracks:
    """A simulator folder of the reference (poses_{1,2}.csv, points_{1,2}.csv, covs_{1,2}.csv: experiments.cc:131-172) as
    a tracks object: what GetFeatures (sim_common.cc:72-107) hands to the solvers -- frame-1 bearings = normalised
    points, frame-2 bearings and covariances through the unscented transform (K^-1 = I, kappa = 1) -- with the start pose
    of sim_common.cc:205-231 (ground truth perturbed by <= 0.01 rad / 0.01, recorded numpy seed instead of mt19937)."""
    import torch

    from . import simulation as sim
    from .io_formats import read_experiments, relative_poses
    from .run_simulation import _matrix_to_quat, perturbed_start
    ex = read_experiments(folder)
    counts = np.array([len(p) for p in ex["points_1"]], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    R_gt, t_gt = relative_poses(ex["poses_1"], ex["poses_2"])
    p1, p2 = np.concatenate(ex["points_1"]), np.concatenate(ex["points_2"])
    c2 = np.concatenate(ex["covs_2"])
    b1 = p1 / np.linalg.norm(p1, axis=1, keepdims=True)
    mu, c2t = torch.from_numpy(p2), torch.from_numpy(c2)
    cov = (sim.unscented_bearing_cov(mu, c2t[:, :2, :2]) if camera == "pinhole" else sim.unscented_bearing_cov_omni(mu, c2t)).numpy()
    b2 = p2 / np.linalg.norm(p2, axis=1, keepdims=True)
    R0, t0 = perturbed_start(R_gt, t_gt, np.random.default_rng(seed), init_scaling)
    ids = np.concatenate([np.arange(n, dtype=np.int64) for n in counts]) if len(counts) else np.zeros(0, dtype=np.int64)
    return Tracks(offsets, b1, b2, cov, _matrix_to_quat(R0), t0, ids1=ids, ids2=ids.copy(),
                  data=f"simulator folder {folder} ({camera})")


def sizes_of(path: str) -> np.ndarray:
    """Pair sizes of a tracks file (what `distributed.partition` balances on)."""
    with np.load(path) as z:
        return np.diff(z["offsets"])


def load_tracks(path: str, first_pair: int = 0, last_pair: int | None = None) -> Tracks:
    """Pairs [first_pair, last_pair) of a tracks file (the whole file by default), offsets rebased."""
    with np.load(path) as z:
        off = z["offsets"].astype(np.int64)
        P = len(off) - 1
        last_pair = P if last_pair is None else last_pair
        if not (0 <= first_pair <= last_pair <= P):
            raise ValueError(f"pair range [{first_pair}, {last_pair}) outside [0, {P})")
        a, b = int(off[first_pair]), int(off[last_pair])
        tr = Tracks(offsets=off[first_pair:last_pair + 1] - a,
                    bvs1=np.ascontiguousarray(z["bvs1"][a:b], dtype=np.float64),
                    bvs2=np.ascontiguousarray(z["bvs2"][a:b], dtype=np.float64),
                    covs=np.ascontiguousarray(z["covs"][a:b], dtype=np.float64),
                    init_q=np.ascontiguousarray(z["init_q"][first_pair:last_pair], dtype=np.float64),
                    init_t=np.ascontiguousarray(z["init_t"][first_pair:last_pair], dtype=np.float64),
                    covs_host=(np.ascontiguousarray(z["covs_host"][a:b], dtype=np.float64)
                               if "covs_host" in z.files else None),
                    sequence=(z["sequence"][first_pair:last_pair] if "sequence" in z.files else None),
                    ids1=(z["ids1"][a:b] if "ids1" in z.files else None),
                    ids2=(z["ids2"][a:b] if "ids2" in z.files else None),
                    data=f"tracks:{path}")
    validate(tr)
    return tr


# ---- synthetic stand-in for "all KITTI 00-10 pairs" ------------------------------------------------
def kitti_all_num_pairs(frames=KITTI_FRAMES) -> int:
    return int(sum(f - 1 for f in frames))


def kitti_all_sequence_ids(frames=KITTI_FRAMES) -> np.ndarray:
    return np.concatenate([np.full(f - 1, s, dtype=np.int32) for s, f in enumerate(frames)])


def _chunk_counts(chunk: int, n: int, mean_corr: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng([seed, chunk, 0xC0])
    c = np.rint(mean_corr + 60.0 * rng.standard_normal(n))
    return np.clip(c, 64, mean_corr + 200).astype(np.int64)


def kitti_all_sizes(mean_corr: int = 500, seed: int = 1, frames=KITTI_FRAMES) -> np.ndarray:
    """Correspondences per pair for the whole synthetic set (cheap; every rank computes all of it)."""
    P = kitti_all_num_pairs(frames)
    return np.concatenate([_chunk_counts(c, min(CHUNK, P - c * CHUNK), mean_corr, seed)
                           for c in range((P + CHUNK - 1) // CHUNK)])


def kitti_all_shard(first_pair: int, last_pair: int, mean_corr: int = 500, seed: int = 1,
                    device="cpu", frames=KITTI_FRAMES, outlier_frac: float = 0.0) -> Tracks:
    """Pairs [first_pair, last_pair) of the synthetic KITTI-00..10-sized set, generated on `device`.
    outlier_frac > 0 replaces that share of frame-2 bearings by random directions in front of the camera
    (gross mismatches, what RANSAC is there for) -- drawn per chunk, so still a function of (seed, pair)."""
    import torch

    from . import simulation as sim
    P = kitti_all_num_pairs(frames)
    if not (0 <= first_pair <= last_pair <= P):
        raise ValueError(f"pair range [{first_pair}, {last_pair}) outside [0, {P})")
    parts = []
    for c in range(first_pair // CHUNK, (max(last_pair, first_pair + 1) - 1) // CHUNK + 1):
        lo, hi = c * CHUNK, min(P, (c + 1) * CHUNK)
        if hi <= first_pair or lo >= last_pair:
            continue
        counts = _chunk_counts(c, hi - lo, mean_corr, seed)
        off, f1, f2, cv, _, _, q, t = sim.generate_kitti_like(hi - lo, mean_corr=mean_corr,
                                                              seed=seed * 100_003 + c, device=device,
                                                              counts=counts)
        if outlier_frac > 0.0:
            g = torch.Generator(device=f1.device)
            g.manual_seed(seed * 7_919 + 31 * c + 5)
            bad = torch.rand(f2.shape[0], device=f1.device, generator=g) < outlier_frac
            rnd = torch.randn(f2.shape[0], 3, dtype=torch.float64, device=f1.device, generator=g)
            rnd[:, 2] = rnd[:, 2].abs() + 1.0
            f2 = torch.where(bad[:, None], rnd / rnd.norm(dim=-1, keepdim=True), f2)
        a, b = max(first_pair, lo) - lo, min(last_pair, hi) - lo
        parts.append((off[a:b + 1] - off[a], f1[off[a]:off[b]], f2[off[a]:off[b]], cv[off[a]:off[b]], q[a:b], t[a:b]))
    dev = torch.device(device)
    if not parts:
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=dev)
        return Tracks(np.zeros(1, dtype=np.int64), z(0, 3), z(0, 3), z(0, 3, 3), z(0, 4), z(0, 3),
                      data="synthetic KITTI-like (no KITTI data in this environment)")
    offsets = np.concatenate([[0], np.cumsum(np.concatenate([np.diff(p[0]) for p in parts]))]).astype(np.int64)
    cat = lambda i: torch.cat([p[i] for p in parts])
    return Tracks(offsets, cat(1), cat(2), cat(3), cat(4), cat(5),
                  sequence=kitti_all_sequence_ids(frames)[first_pair:last_pair],
                  data="synthetic KITTI-like (no KITTI data in this environment): sequence lengths of "
                       "KITTI odometry 00-10, forward motion, fx=718.856, ragged track counts")


def main(argv=None) -> int:
    """python -m pnec_amd.tracks check f.npz [--min-corr N]  |  from-experiments <folder> out.npz [--camera pinhole|omni]"""
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="python -m pnec_amd.tracks", description=main.__doc__)
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("check", help="validate a tracks file (shapes, unit bearings, PSD covariances, matched ids, counts)")
    c.add_argument("path")
    c.add_argument("--min-corr", type=int, default=10)
    f = sub.add_parser("from-experiments", help="convert a simulator CSV folder of the reference into a tracks file")
    f.add_argument("folder")
    f.add_argument("out")
    f.add_argument("--camera", choices=("pinhole", "omni"), default="pinhole")
    f.add_argument("--seed", type=int, default=1)
    a = ap.parse_args(argv)
    if a.cmd == "check":
        try:
            rep = check(load_tracks(a.path), a.min_corr)
        except (ValueError, KeyError, OSError) as e:
            rep = {"ok": False, "problems": [f"{type(e).__name__}: {e}"]}
        print(json.dumps(rep, indent=1))
        return 0 if rep["ok"] else 1
    tr = from_experiments(a.folder, a.camera, seed=a.seed)
    save_tracks(a.out, tr)
    print(json.dumps(check(tr), indent=1))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
