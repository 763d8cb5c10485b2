"""The per-frame handle (pnec_hip_frame_*): PNEC::Solve for one frame pair per call on persistent resources,
and the capacity-shaped batch it is built on (pnec_hip_problem_create_capacity / _reshape)."""
import ctypes as C
import math

import numpy as np
import pytest

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd.frame import FrameSolver

pytestmark = pytest.mark.gpu


def _pair(n, seed, outliers=0.12):
    g = sim.generate(1, max(n, 1), seed=seed)
    b1, b2, cv = g.bvs1[0, :n].numpy().copy(), g.bvs2[0, :n].numpy().copy(), g.covs2[0, :n].numpy().copy()
    rng = np.random.default_rng(seed)
    k = int(n * outliers)
    if k:
        bad = rng.choice(n, k, replace=False)
        v = rng.normal(size=(k, 3))
        b2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    return b1, b2, cv, g.init_q[0].numpy(), g.init_t[0].numpy()


def _batch_chain(b1, b2, cv, q0, t0, opts):
    n = len(b1)
    with Batch(capi.MODE_TARGET, np.array([0, n], dtype=np.int64)) as b:
        if n:
            b.fill(b1, b2, cv)
        q, t, mask, cnt = b.solve_pipeline(q0[None], t0[None], options=opts, want_inliers=True)
    return q[0], t[0], mask.astype(bool), int(cnt[0])


def test_frame_solve_is_bit_identical_to_the_one_pair_batch_call_for_sizes_that_grow_and_shrink(oracle):
    """One handle, a sequence of frames of different sizes (shrinking, growing, empty, beyond one wavefront's 512,
    beyond 1024), every Options branch of PNEC::Solve: pose, inlier mask and count equal the batch path's
    (pnec_hip_problem_create + fill + pnec_hip_solve_pipeline on a one-pair batch) BIT FOR BIT -- the handle
    re-shapes one persistent batch and reads the arrays zero-copy, the launches are the same -- and the default
    branch also matches the oracle's chain."""
    sizes = [512, 100, 700, 0, 33, 1500, 64, 511, 2048, 9]
    branches = [dict(), dict(use_ransac=0), dict(use_nec=1), dict(weighted_iterations=1), dict(use_ceres=0),
                dict(weighted_iterations=0), dict(use_nec=1, use_ransac=0, use_ceres=0)]
    with FrameSolver(max_corr=2048) as fs:
        assert fs.capacity == 2048
        for i, n in enumerate(sizes):
            b1, b2, cv, q0, t0 = _pair(n, 100 + i)
            for kw in (branches if i < 4 else branches[:1] + [branches[1 + i % 6]]):
                opts = capi.default_pipeline_options(**kw)
                q, t, mask, cnt = fs.solve(b1, b2, cv, q0, t0, options=opts)
                if n == 0:
                    continue                                  # nothing to compare: the call must just survive
                rq, rt, rmask, rcnt = _batch_chain(b1, b2, cv, q0, t0, opts)
                assert np.array_equal(q, rq) and np.array_equal(t, rt), (n, kw, q - rq)
                assert np.array_equal(mask, rmask) and cnt == rcnt, (n, kw)
                if not kw.get("use_ransac", 1):
                    assert cnt == 0 and not mask.any()
            if n >= 64:   # the default branch against the oracle's chain (pair_id 0: what a single Solve draws)
                q, t, mask, cnt = fs.solve(b1, b2, cv, q0, t0)
                o = oracle.solve_chain_batch(np.array([0, n]), b1, b2, cv, q0[None])
                assert np.array_equal(mask, o["mask"]) and cnt == int(o["inlier_count"][0])
                assert math.radians(oracle.rotational_difference_deg(oracle.rot_from_quat(q), oracle.rot_from_quat(o["q"][0]))) <= 1e-6
        with pytest.raises(capi.PnecHipError):               # beyond the capacity: refused, handle still usable
            fs.solve(*_pair(2049, 7))
        b1, b2, cv, q0, t0 = _pair(300, 8)
        q, t, mask, cnt = fs.solve(b1, b2, cv, q0, t0)
        rq, rt, rmask, rcnt = _batch_chain(b1, b2, cv, q0, t0, None)
        assert np.array_equal(q, rq) and np.array_equal(mask, rmask)
        with pytest.raises(capi.PnecHipError):               # the PNEC chain needs covariances
            fs.solve(b1, b2, None, q0, t0)
        fs.solve(b1, b2, None, q0, t0, options=capi.default_pipeline_options(use_nec=1))   # the NEC chain does not


def test_capacity_shaped_batch_reshapes_without_reallocating_and_matches_fresh_batches():
    """pnec_hip_problem_create_capacity + pnec_hip_problem_reshape: one batch takes a sequence of ragged shapes
    (more pairs, fewer pairs, larger and smaller pairs); after each reshape + fill its refinement equals a freshly
    created batch's bit for bit; shapes beyond the capacity are refused and leave the batch as it was."""
    import torch
    L = capi.lib()
    rng = np.random.default_rng(3)
    h = C.c_void_p()
    capi.check(L.pnec_hip_problem_create_capacity(0, capi.MODE_TARGET, 40, 12000, C.byref(h)))
    try:
        assert L.pnec_hip_problem_num_pairs(h) == 0
        for trial in range(6):
            P = int(rng.integers(1, 41))
            counts = rng.integers(1, 12000 // P + 1, size=P).astype(np.int64)
            off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            g = sim.generate(P, int(counts.max()), seed=40 + trial)
            f1 = np.concatenate([g.bvs1[p, :n].numpy() for p, n in enumerate(counts)])
            f2 = np.concatenate([g.bvs2[p, :n].numpy() for p, n in enumerate(counts)])
            cv = np.concatenate([g.covs2[p, :n].numpy() for p, n in enumerate(counts)])
            capi.check(L.pnec_hip_problem_reshape(h, P, off.ctypes.data, None))
            assert L.pnec_hip_problem_num_pairs(h) == P and L.pnec_hip_problem_num_correspondences(h) == int(off[-1])
            view = Batch.__new__(Batch)                      # the Batch front-end over the same handle (not owning it)
            view._lib, view.mode, view.device, view._h, view.n_pairs, view._offsets = L, capi.MODE_TARGET, 0, h, P, off
            view.fill(f1, f2, cv)
            got = view.solve(g.init_q.numpy(), g.init_t.numpy())
            view._h = None
            with Batch(capi.MODE_TARGET, off) as fresh:
                fresh.fill(f1, f2, cv)
                want = fresh.solve(g.init_q.numpy(), g.init_t.numpy())
            assert np.array_equal(got.q, want.q) and np.array_equal(got.t, want.t), trial
            assert np.array_equal(got.iterations, want.iterations) and np.array_equal(got.cost, want.cost)
        too_many = np.arange(42, dtype=np.int64)
        assert L.pnec_hip_problem_reshape(h, 41, too_many.ctypes.data, None) == -1
        too_big = np.array([0, 20000], dtype=np.int64)       # > 12000 + 63 * 40 (padding room)
        assert L.pnec_hip_problem_reshape(h, 1, too_big.ctypes.data, None) == -1
        assert L.pnec_hip_problem_num_pairs(h) == P          # unchanged by the refused shapes
        fixed = C.c_void_p()
        capi.check(L.pnec_hip_problem_create(0, capi.MODE_TARGET, 1, too_big.ctypes.data, C.byref(fixed)))
        assert L.pnec_hip_problem_reshape(fixed, 1, too_big.ctypes.data, None) == -1   # not capacity-shaped
        L.pnec_hip_problem_destroy(fixed)
    finally:
        L.pnec_hip_problem_destroy(h)
    torch.cuda.synchronize()


def test_sequences_in_lockstep_with_device_chained_starts_equal_one_call_per_frame():
    """bench.py's `kitti_all_sequences_in_lockstep`: the sequences of the KITTI-like set advance side by side, one batched
    whole-chain call per time step on a capacity batch that is re-shaped and re-filled per step (Batch.with_capacity /
    reshape), every start pose the previous step's OUTPUT TENSOR of the same sequence -- nothing returns to the host
    between steps.  A short version (sequence lengths / 12): every pose of the first 12 steps equals, bit for bit, the
    same frame solved alone through pnec_hip_frame_solve with the same start pose and RANSAC pair id; all poses finite,
    every sequence ran to its end."""
    import torch
    import bench
    from pnec_amd import capi
    from pnec_amd import tracks as tk
    r = bench.lockstep_sequences(torch.device("cuda:0"), capi, tk, quick=True, check_steps=12)
    assert r["bitwise_equal_to_one_call_per_frame"] and r["n_compared"] >= 12 * 9
    assert r["sequences"] == 11 and r["pairs"] == sum(max(2, f // 12) - 1 for f in tk.KITTI_FRAMES)
    q = r["q"]
    assert bool(torch.isfinite(q).all()) and float(((q * q).sum(-1) - 1).abs().max()) < 1e-12
