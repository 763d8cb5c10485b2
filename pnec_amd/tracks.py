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
