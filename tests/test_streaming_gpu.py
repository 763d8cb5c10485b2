"""The per-frame call pattern on the GPU: the streaming handle (pnec_hip_stream_*) and the device-resident
PNEC::Solve chain (pnec_hip_solve_pipeline).  Reference call sites: Frame2Frame::PNECAlign -> PNEC::Solve
once per frame (src/rel_pose_estimation/frame2frame.cc:122-141), PNECCeres::Optimize once per pybind call
(python/pypnec.cpp:55-65)."""
import math

import numpy as np
import pytest
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd.streaming import Stream

pytestmark = pytest.mark.gpu


def _quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_streamed_kitti_like_sequence_is_bit_identical_to_the_batched_call():
    """BASELINE config 3: ~4.5k consecutive frame pairs streamed on one GPU, ONE PAIR PER CALL as the
    odometry does, against the same pairs solved as one ragged batch: every output bit for bit."""
    P = 4541
    offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=500, seed=3)
    f1, f2, c2, q0, t0 = (x.numpy() for x in (f1, f2, c2, q0, t0))
    with Batch(capi.MODE_TARGET, offsets) as b:
        b.fill(f1, f2, c2)
        ref = b.solve(q0, t0)
    got = {k: [] for k in ("q", "t", "cost", "iterations", "status")}
    with Stream(max_corr=int(np.diff(offsets).max()), slots=8) as st:
        tickets = []
        for p in range(P):                       # a window of 8 frames in flight
            a, e = offsets[p], offsets[p + 1]
            tickets.append(st.submit(capi.MODE_TARGET, f1[a:e], f2[a:e], c2[a:e], None, q0[p], t0[p]))
            if len(tickets) == 8:
                r = st.wait(tickets.pop(0))
                for k in got:
                    got[k].append(getattr(r, k)[0])
        while tickets:
            r = st.wait(tickets.pop(0))
            for k in got:
                got[k].append(getattr(r, k)[0])
    np.testing.assert_array_equal(np.stack(got["q"]), ref.q)
    np.testing.assert_array_equal(np.stack(got["t"]), ref.t)
    np.testing.assert_array_equal(np.array(got["cost"]), ref.cost)
    np.testing.assert_array_equal(np.array(got["iterations"]), ref.iterations)
    np.testing.assert_array_equal(np.array(got["status"]), ref.status)


@pytest.mark.parametrize("mode", [capi.MODE_NEC, capi.MODE_TARGET, capi.MODE_HOST, capi.MODE_SYM])
def test_stream_families_sizes_and_multi_pair_submits_match_the_batch_path(mode, oracle):
    """every residual family, sizes across the geometry ladder (incl. empty, > 4096 = the staged route),
    several pairs per submit, poll before wait, options honoured -- same bits as the batch path, and the
    oracle within the north-star tolerance"""
    rng = np.random.default_rng(5)
    sizes = [0, 1, 9, 64, 65, 100, 256, 300, 512, 513, 1024, 2048, 2049]
    if mode != capi.MODE_SYM:
        sizes += [4096]
    g = sim.generate(len(sizes), max(sizes), seed=77)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    f1 = np.concatenate([g.bvs1[p].numpy()[:n] for p, n in enumerate(sizes)])
    f2 = np.concatenate([g.bvs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    S2 = np.concatenate([g.covs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    c2 = None if mode == capi.MODE_NEC else S2
    c1 = np.roll(S2, 1, axis=0) * 0.8 if mode == capi.MODE_SYM else None
    q0, t0 = g.init_q.numpy(), g.init_t.numpy()
    opts = capi.default_options(max_num_iterations=7, check_convergence=int(rng.integers(0, 2)))
    with Batch(mode, offsets) as b:
        b.fill(f1, f2, c2, c1)
        ref = b.solve(q0, t0, reg=1e-13, options=opts)
    with Stream(max_corr=int(offsets[-1]), max_pairs=len(sizes), slots=3) as st:
        # (a) one pair per submit
        for p, n in enumerate(sizes):
            a, e = offsets[p], offsets[p + 1]
            tk = st.submit(mode, f1[a:e], f2[a:e], None if c2 is None else c2[a:e], None if c1 is None else c1[a:e],
                           q0[p], t0[p], 1e-13, opts)
            r = st.wait(tk)
            np.testing.assert_array_equal(r.q[0], ref.q[p])
            np.testing.assert_array_equal(r.t[0], ref.t[p])
            assert r.cost[0] == ref.cost[p] or (np.isnan(r.cost[0]) and np.isnan(ref.cost[p]))
            assert r.iterations[0] == ref.iterations[p] and r.status[0] == ref.status[p]
        # (b) all pairs in ONE submit (one launch per geometry in use), polled
        tk = st.submit(mode, f1, f2, c2, c1, q0, t0, 1e-13, opts, offsets=offsets)
        while not st.poll(tk):
            pass
        r = st.wait(tk)
        np.testing.assert_array_equal(r.q, ref.q)
        np.testing.assert_array_equal(r.iterations, ref.iterations)
        # (c) a pair beyond the register-resident geometries takes the staged route
        big = 5000 if mode != capi.MODE_SYM else 2100
        gb = sim.generate(1, big, seed=78)
        Sb = gb.covs2[0].numpy()
        cb2 = None if mode == capi.MODE_NEC else Sb
        cb1 = np.roll(Sb, 1, axis=0) * 0.8 if mode == capi.MODE_SYM else None
        with Stream(max_corr=big, slots=2) as st2:
            rb = st2.solve(mode, gb.bvs1[0].numpy(), gb.bvs2[0].numpy(), cb2, cb1, gb.init_q[0].numpy(), gb.init_t[0].numpy())
        s = oracle.solve(mode, gb.bvs1[0].numpy(), gb.bvs2[0].numpy(), cb2, cb1, 1e-13 if mode != capi.MODE_NEC else 0.0,
                         gb.init_q[0].numpy(), gb.init_t[0].numpy(), oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(_quat_to_R(rb.q[0]), s.R)) <= 1e-6
        assert rb.iterations[0] == s.iterations
    # the oracle on a few of the streamed pairs
    oo = oracle.default_options(max_num_iterations=7, check_convergence=opts.check_convergence)
    for p in (4, 8, 10):
        a, e = offsets[p], offsets[p + 1]
        s = oracle.solve(mode, f1[a:e], f2[a:e], None if c2 is None else c2[a:e], None if c1 is None else c1[a:e],
                         1e-13, q0[p], t0[p], oo)
        assert math.radians(oracle.rotational_difference_deg(_quat_to_R(ref.q[p]), s.R)) <= 1e-6


def test_stream_argument_errors_and_ticket_rules():
    g = sim.generate(1, 32, seed=1)
    f1, f2, c2, q, t = (x[0].numpy() for x in (g.bvs1, g.bvs2, g.covs2, g.init_q, g.init_t))
    with Stream(max_corr=16, slots=2) as st:
        with pytest.raises(capi.PnecHipError, match="more correspondences"):
            st.submit(capi.MODE_TARGET, f1, f2, c2, None, q, t)
        with pytest.raises(capi.PnecHipError):           # TARGET needs covariances
            st.submit(capi.MODE_TARGET, f1[:8], f2[:8], None, None, q, t)
        tk = st.submit(capi.MODE_NEC, f1[:16], f2[:16], None, None, q, t)
        r = st.wait(tk)
        assert np.isfinite(r.q).all()
        st._pairs[tk] = 1
        with pytest.raises(capi.PnecHipError, match="ticket"):   # a ticket is collected once
            st.wait(tk)
        # a full ring is an error, not a silent drop of the oldest result: slots = 2 -> the third submit must wait
        t1 = st.submit(capi.MODE_NEC, f1[:16], f2[:16], None, None, q, t)
        t2 = st.submit(capi.MODE_NEC, f1[:16], f2[:16], None, None, q, t)
        with pytest.raises(capi.PnecHipError, match="uncollected"):
            st.submit(capi.MODE_NEC, f1[:16], f2[:16], None, None, q, t)
        r1 = st.wait(t1)
        t3 = st.submit(capi.MODE_NEC, f1[:16], f2[:16], None, None, q, t)
        r2, r3 = st.wait(t2), st.wait(t3)
        np.testing.assert_array_equal(r1.q, r.q)
        np.testing.assert_array_equal(r2.q, r.q)
        np.testing.assert_array_equal(r3.q, r.q)
    with pytest.raises(capi.PnecHipError):
        Stream(max_corr=0)


def test_pipeline_equals_the_stage_by_stage_chain_and_never_needs_host_sizes(oracle):
    """pnec_hip_solve_pipeline = RANSAC eigensolver -> InlierExtraction -> weighted eigensolver -> refinement
    chained on the device: same bits as the stage-by-stage calls, in both memory spaces, for every Options
    branch PNEC::Solve has; the inlier batch's sizes are only fetched when asked for"""
    sizes = [300, 512, 128, 8, 40, 0, 700]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    B = len(sizes)
    g = sim.generate(B, 700, seed=95)
    rng = np.random.default_rng(2)
    f1 = np.concatenate([g.bvs1[p].numpy()[:n] for p, n in enumerate(sizes)])
    f2 = np.concatenate([g.bvs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    c2 = np.concatenate([g.covs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    for p in range(B):
        sl = np.arange(offsets[p], offsets[p + 1])
        if len(sl) >= 20:
            bad = rng.choice(sl, len(sl) // 5, replace=False)
            v = rng.normal(size=(len(bad), 3))
            f2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    q0, t0 = g.init_q.numpy(), g.init_t.numpy()
    with Batch(capi.MODE_TARGET, offsets) as b:
        b.fill(f1, f2, c2)
        # stage by stage (host space)
        qr, tr, mask, cnt, its = b.ransac_eigensolver(q0, seed=1)
        sel = b.select(mask)
        qw, tw = sel.weighted_eigensolver(qr, tr, 1e-13, 10)
        want = sel.solve(qw, tw)
        np.testing.assert_array_equal(np.diff(sel.offsets), cnt)      # sizes fetched on demand
        assert sel.num_correspondences == int(mask.sum())
        # the chain, host space
        q, t, m2, c2n = b.solve_pipeline(q0, t0, want_inliers=True)
        np.testing.assert_array_equal(m2, mask)
        np.testing.assert_array_equal(c2n, cnt)
        np.testing.assert_array_equal(q, want.q)
        np.testing.assert_array_equal(t, want.t)
        # device space (asynchronous), twice on the same batch (cached inlier batch re-used)
        for _ in range(2):
            qd, td, md, cd = b.solve_pipeline(torch.from_numpy(q0).cuda(), torch.from_numpy(t0).cuda(), want_inliers=True)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(qd.cpu().numpy(), want.q)
            np.testing.assert_array_equal(md.cpu().numpy(), mask)
        # Options branches
        o = capi.default_pipeline_options(use_ransac=0)
        qn, tn = b.nec_eigensolver(q0)
        qw2, tw2 = b.weighted_eigensolver(qn, tn, 1e-13, 10)
        w2 = b.solve(qw2, tw2)
        q, t = b.solve_pipeline(q0, t0, o)
        np.testing.assert_array_equal(q, w2.q)
        o = capi.default_pipeline_options(use_ceres=0)
        q, t = b.solve_pipeline(q0, t0, o)
        np.testing.assert_array_equal(q, qw)
        np.testing.assert_array_equal(t, tw)
        o = capi.default_pipeline_options(weighted_iterations=1)
        q, t = b.solve_pipeline(q0, t0, o)
        np.testing.assert_array_equal(q, sel.solve(qr, tr).q)
        o = capi.default_pipeline_options(weighted_iterations=0, use_ransac=0)
        q, t = b.solve_pipeline(q0, t0, o)
        np.testing.assert_array_equal(q, b.solve(q0, t0).q)
        o = capi.default_pipeline_options(use_nec=1)
        q, t = b.solve_pipeline(q0, t0, o)
        with Batch(capi.MODE_NEC, sel.offsets) as nb:           # NECCeresSolver on the inlier bearings
            keep = mask.astype(bool)
            nb.fill(f1[keep], f2[keep])
            wn = nb.solve(qr, tr, reg=0.0)
        for p in range(B):
            assert math.radians(oracle.rotational_difference_deg(_quat_to_R(q[p]), _quat_to_R(wn.q[p]))) <= 1e-9
        o = capi.default_pipeline_options(use_nec=1, use_ceres=0)
        q, t = b.solve_pipeline(q0, t0, o)
        np.testing.assert_array_equal(q, qr)
        with pytest.raises(capi.PnecHipError):
            b.solve_pipeline(q0, t0, capi.default_pipeline_options(ransac_sample_size=17))
        sel.close()
    # against the oracle's chain on the pairs RANSAC can work with
    for p in (0, 1, 2, 6):
        sl = slice(offsets[p], offsets[p + 1])
        Ro, to, mo, _ = oracle.ransac_eigensolver(f1[sl], f2[sl], g.init_R[p].numpy(), seed=1, pair_id=p)
        Rw, tw_ = oracle.weighted_eigensolver(f1[sl][mo], f2[sl][mo], c2[sl][mo], Ro, to, 1e-13, 10)
        s = oracle.solve(oracle.MODE_TARGET, f1[sl][mo], f2[sl][mo], c2[sl][mo], None, 1e-13, oracle.quat_from_rot(Rw),
                         tw_, oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(_quat_to_R(want.q[p]), s.R)) <= 1e-6


@pytest.mark.parametrize("mode", [capi.MODE_NEC, capi.MODE_TARGET, capi.MODE_SYM])
def test_fused_keypoint_ingest_is_bitwise_unscented_transform_plus_fill(mode, oracle):
    """pnec_hip_problem_fill_keypoints (KeyPoint::Unproject for both frames: keypoints.cc:49-62) against the
    two-step path it fuses -- pnec_hip_unscented_transform, then pnec_hip_problem_fill -- plane for plane,
    bit for bit, in both memory spaces; and against the oracle's UnscentedTransform."""
    from pnec_amd import frontend
    rng = np.random.default_rng(31)
    sizes = [100, 1, 0, 513, 64]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    M = int(offsets[-1])
    K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1.0]])
    Kinv = np.linalg.inv(K)
    p1 = np.stack([rng.uniform(0, 1241, M), rng.uniform(0, 376, M)], 1)
    p2 = p1 + rng.normal(size=(M, 2)) * 5

    def cov2x2():
        A = rng.normal(size=(M, 2, 2)) * 0.4
        return A @ np.transpose(A, (0, 2, 1)) + 0.02 * np.eye(2)
    c2, c1 = cov2x2(), cov2x2()
    mu = lambda p: np.concatenate([p, np.ones((M, 1))], 1)
    c33 = lambda c: np.pad(c, ((0, 0), (0, 1), (0, 1)))
    b2, S2 = frontend.unscented_transform(mu(p2), c33(c2), Kinv, 1.0, frontend.CAMERA_PINHOLE)
    b1, S1 = frontend.unscented_transform(mu(p1), c33(c1), Kinv, 1.0, frontend.CAMERA_PINHOLE)
    np.testing.assert_allclose(b1, (mu(p1) @ Kinv.T) / np.linalg.norm(mu(p1) @ Kinv.T, axis=1, keepdims=True), atol=1e-15)
    for i in range(0, M, 41):
        np.testing.assert_allclose(S2[i], oracle.unscented_transform(mu(p2)[i], c33(c2)[i], Kinv, 1.0, frontend.CAMERA_PINHOLE),
                                   rtol=1e-9, atol=1e-22)
    cov_args = {capi.MODE_NEC: (None, None), capi.MODE_TARGET: (S2, None), capi.MODE_SYM: (S2, S1)}[mode]
    kp_args = {capi.MODE_NEC: (None, None), capi.MODE_TARGET: (c2, None), capi.MODE_SYM: (c2, c1)}[mode]
    with Batch(mode, offsets) as two_step, Batch(mode, offsets) as fused, Batch(mode, offsets) as fused_dev:
        two_step.fill(b1, b2, *cov_args)
        want = two_step.export_payload()
        fused.fill_keypoints(p1, p2, *kp_args, K_inv=Kinv)
        np.testing.assert_array_equal(fused.export_payload(), want)
        t = lambda a: None if a is None else torch.from_numpy(a).cuda()
        fused_dev.fill_keypoints(t(p1), t(p2), t(kp_args[0]), t(kp_args[1]), K_inv=Kinv)
        np.testing.assert_array_equal(fused_dev.export_payload(), want)
        # partial re-fill of a pair range
        fused.fill_keypoints(p1[offsets[3]:], p2[offsets[3]:], *(None if a is None else a[offsets[3]:] for a in kp_args),
                             K_inv=Kinv, first_pair=3, n_pairs=2)
        np.testing.assert_array_equal(fused.export_payload(), want)
        with pytest.raises(ValueError):
            fused.fill_keypoints(p1[:5], p2, *kp_args, K_inv=Kinv)
        k9 = np.ascontiguousarray(Kinv.T.reshape(9))
        with pytest.raises(capi.PnecHipError, match="pinhole"):   # KeyPoint::Unproject is pinhole only
            capi.check(capi.lib().pnec_hip_problem_fill_keypoints(fused._h, 0, 1, p1.ctypes.data, p2.ctypes.data, None, None,
                                                                  k9.ctypes.data, 1.0, 0, capi.MEM_HOST, None))


def test_loading_the_library_before_torch_keeps_one_hip_runtime():
    """`g.build(); g.smoke()` in one process loads libpnec_hip.so before anything touched torch: the binding
    must make PyTorch's HIP runtime the process's only one (it once ended in 'hipSetDevice failed')."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from pnec_amd import capi\n"
            "L = capi.lib()\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "capi.check(L.pnec_hip_selftest(0))\n"
            "x = torch.ones(4, device='cuda:0', dtype=torch.float64)\n"
            "print('sum', float(x.sum()))\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "sum 4.0" in out.stdout


def test_pipeline_on_the_kitti_like_stream_matches_the_oracle_chain(oracle):
    """BASELINE configs 3 / 5 data regime (forward motion, low parallax, ragged sizes) with 10 % gross
    outliers: the one-call chain against the oracle's RANSAC -> weighted eigensolver + SCF -> refinement,
    pair by pair: same inlier masks, same RANSAC iteration counts, rotations within the north-star tolerance
    (tools/verify_pipeline_kitti.py runs 3 000 pairs: masks identical, max 8.4e-13 rad)"""
    P = 24
    offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=400, seed=21)
    gen = torch.Generator().manual_seed(3)
    M = f1.shape[0]
    bad = torch.rand(M, generator=gen) < 0.10
    rnd = torch.randn(M, 3, dtype=torch.float64, generator=gen)
    rnd[:, 2] = rnd[:, 2].abs() + 1.0
    f2 = torch.where(bad[:, None], rnd / rnd.norm(dim=-1, keepdim=True), f2)
    f1, f2, c2, q0, t0 = (x.numpy() for x in (f1, f2, c2, q0, t0))
    off = np.asarray(offsets)
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, c2)
        q, t, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
        _, _, _, _, its = b.ransac_eigensolver(q0, seed=1)
    mask = np.asarray(mask).astype(bool)
    for p in range(P):
        a, e = off[p], off[p + 1]
        R0 = _quat_to_R(q0[p])
        Rr, trr, m, it = oracle.ransac_eigensolver(f1[a:e], f2[a:e], R0, seed=1, pair_id=p)
        np.testing.assert_array_equal(m, mask[a:e])
        assert int(it) == int(its[p]) and int(m.sum()) == int(cnt[p])
        Rw, tw = oracle.weighted_eigensolver(f1[a:e][m], f2[a:e][m], c2[a:e][m], Rr, trr)
        s = oracle.solve(capi.MODE_TARGET, f1[a:e][m], f2[a:e][m], c2[a:e][m], None, 1e-13, oracle.quat_from_rot(Rw), tw,
                         oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(_quat_to_R(q[p]), s.R)) <= 1e-6, p


def test_pipeline_survives_degenerate_pairs():
    """pure rotation (no translation: a double zero eigenvalue), identical bearings, every correspondence the same
    point, a NaN bearing, zero covariances, one covariance 1e12 times the others -- inside an ordinary batch, through
    the whole chain in one call: it ends, the ordinary pairs are untouched by their neighbours, nothing but the NaN
    pair may come out non-finite"""
    P, N = 16, 512
    g = sim.generate(P, N, seed=5)
    f1, f2, c2 = g.bvs1.clone(), g.bvs2.clone(), g.covs2.clone()
    f2[0] = (g.R_gt[0].T @ f1[0].T).T
    f2[1] = f1[1]
    f1[2] = f1[2][0:1].expand(N, 3).clone()
    f2[2] = f2[2][0:1].expand(N, 3).clone()
    f1[4, 7, 0] = float("nan")
    c2[5] = 0.0
    c2[6, 3] = c2[6, 3] * 1e12
    q0, t0 = g.init_q.numpy(), g.init_t.numpy()
    args = lambda a, b_, c: (a.reshape(-1, 3).numpy(), b_.reshape(-1, 3).numpy(), c.reshape(-1, 3, 3).numpy())
    with Batch.uniform(capi.MODE_TARGET, P, N) as b:
        b.fill(*args(f1, f2, c2))
        q, t, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
        qn, tn = b.nec_eigensolver(q0)
    with Batch.uniform(capi.MODE_TARGET, P, N) as b:          # the same batch without the troublemakers
        b.fill(*args(g.bvs1, g.bvs2, g.covs2))
        q_ref, t_ref, _, cnt_ref = b.solve_pipeline(q0, t0, want_inliers=True)
    q, t, qn = np.asarray(q), np.asarray(t), np.asarray(qn)
    for p in range(P):
        if p == 4:
            continue
        assert np.isfinite(q[p]).all() and np.isfinite(t[p]).all() and np.isfinite(qn[p]).all(), p
    np.testing.assert_array_equal(q[7:], np.asarray(q_ref)[7:])      # pairs are independent
    np.testing.assert_array_equal(np.asarray(cnt)[7:], np.asarray(cnt_ref)[7:])


def test_select_view_is_bitwise_select_and_reuses_one_target():
    """pnec_hip_problem_select_view (round 4): InlierExtraction into the batch's CACHED target -- what the timed
    PNEC::Solve overloads use per frame instead of creating and destroying a batch.  Same planes, counts and offsets
    as pnec_hip_problem_select, for host and device masks, twice in a row with different masks (the second call
    replaces the first result in place), and the later stages give the same bits on either."""
    import torch
    counts = np.array([100, 64, 513, 9, 300], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(counts)])
    g = sim.generate(1, int(off[-1]), seed=77, device="cuda:0")
    poses = sim.generate(len(counts), 4, seed=78, device="cuda:0")
    f1, f2, cv = g.bvs1[0], g.bvs2[0], g.covs2[0]
    rng = np.random.default_rng(5)
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, cv)
        for rep in range(2):
            mask = (rng.random(int(off[-1])) < (0.8 if rep == 0 else 0.5)).astype(np.uint8)
            for m in (mask, torch.from_numpy(mask).cuda()):
                own = b.select(m)
                view = b.select(m, view=True)
                np.testing.assert_array_equal(view.offsets, own.offsets)
                # the planes: a target keeps the SOURCE's block layout; pair p's kept correspondences fill the first
                # round_up(m_p, 64) doubles of each of its 12 planes (what lies beyond belongs to nobody)
                pv, po = view.export_payload(), own.export_payload()
                block = np.concatenate([[0], np.cumsum(12 * ((counts + 63) // 64 * 64))])
                kept = np.diff(own.offsets)
                for p in range(len(counts)):
                    used = 12 * int((kept[p] + 63) // 64 * 64)
                    np.testing.assert_array_equal(pv[block[p]:block[p] + used], po[block[p]:block[p] + used])
                r_own = own.solve(poses.init_q, poses.init_t)
                r_view = view.solve(poses.init_q, poses.init_t)
                assert torch.equal(r_own.q, r_view.q) and torch.equal(r_own.iterations, r_view.iterations)
                view.close()          # (a no-op for the library: the target belongs to b)
                own.close()
