"""Whole-chain parity AT SCALE, in the suite the driver runs: PNEC::Solve with the reference's default Options
(pnec.cc:77-124: RANSAC eigensolver -> InlierExtraction -> WeightedEigensolver + SCF -> CeresSolver) as ONE
pnec_hip_solve_pipeline call over 20 000 pairs x 512 correspondences with 10 % gross mismatches, against the
oracle's chain (oracle/pnec_oracle_frontend.c: pnec_oracle_solve_chain_batch, OpenMP over all host cores) on the
same inputs and the same counter-based draws.

What is asserted, and why it is not simply "every pair <= 1e-6 rad": two of the reference's own algorithms are
discontinuous in their inputs at rounding level -- a RANSAC hypothesis whose 10-point eigenvalue minimisation has
two local minima (which one an iteration ends in is decided by the last bits of an Armijo test), and Ceres'
accept / reject / stop sequence on an ill-conditioned refinement (30-50 LM iterations).  Over 100 000 pairs 6-9
pairs end beyond 1e-6 rad for those reasons.  So: masks identical for >= 99.99 % of the pairs,
99th percentile <= 1e-10 rad, at most 3 pairs beyond 1e-6 rad -- and every one of those must be EXPLAINED in the
test: the oracle's later stages, started from the DEVICE's own intermediate (its inlier mask and eigensolver pose,
or its weighted-stage pose), must reproduce the device's final pose to <= 1e-6 rad, i.e. the stage that follows
agrees and the difference is the upstream decision alone."""
import json
import os

import numpy as np
import pytest
import torch

from pnec_amd import Batch, capi
from pnec_amd import simulation as sim

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-6   # rad: BASELINE.json north_star


def _angles(q_a, q_b):
    """rotation angle between unit quaternions (xyzw), small-angle safe, vectorised"""
    a, b = np.asarray(q_a), np.asarray(q_b)
    d = np.abs(np.sum(a * b, axis=1)).clip(0, 1)
    v = np.stack([a[:, 3] * b[:, 0] - a[:, 0] * b[:, 3] - a[:, 1] * b[:, 2] + a[:, 2] * b[:, 1],
                  a[:, 3] * b[:, 1] + a[:, 0] * b[:, 2] - a[:, 1] * b[:, 3] - a[:, 2] * b[:, 0],
                  a[:, 3] * b[:, 2] - a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0] - a[:, 2] * b[:, 3]], 1)
    return 2.0 * np.arctan2(np.linalg.norm(v, axis=1), d)


def _check_chain_against_oracle(oracle, es_scheme, off, f1, f2, cv, q0, t0, label):
    """The one-call chain on the device under `es_scheme` against the checker's chain under the same scheme, over the
    pairs (off, f1, f2, cv: device tensors; q0, t0 start poses).  Scheme 0 (damped Newton): every pair identical up to a
    handful that are re-derived from the device's own intermediates.  Scheme 2 (the facade's default: MINPACK-style LM on
    the reduced-Cayley gradient): the same, except that the iteration ITSELF stalls on a flat valley of |grad| where M's
    two smallest eigenvalues lie close (about 1 % of pairs) and ends at a point its rounding picks -- such a pair must be
    one where the CHECKER's own minimisation ended in a stall (MINPACK info 1) or at maxfev (5), or where the device's
    result is not a stationary point either: explained, not tolerated blindly (the rule of
    tests/test_opengv_schemes.py::test_whole_chain_device_vs_checker_2000_pairs, here at scale)."""
    cores = oracle.usable_threads()
    P = len(off) - 1
    po = capi.default_pipeline_options(eigensolver_scheme=es_scheme)
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, cv)
        q, t, mask, cnt = b.solve_pipeline(q0, t0, options=po, want_inliers=True)          # the product call
        # the same chain stage by stage (bit-identical by construction; its intermediates explain outliers)
        b.set_eigensolver_scheme(es_scheme)
        qr, tr, mask_s, cnt_s, its = b.ransac_eigensolver(q0, seed=1)
        sel = b.select(mask_s)
        qw, tw = sel.weighted_eigensolver(qr, tr, 1e-13, 10)
        res = sel.solve(qw, tw)
        sel.close()
    torch.cuda.synchronize()
    assert torch.equal(q, res.q) and torch.equal(t, res.t) and torch.equal(mask, mask_s) and torch.equal(cnt, cnt_s)
    gq, gmask = q.cpu().numpy(), mask.cpu().numpy().astype(bool)
    f1, f2, cv = (x.cpu().numpy() for x in (f1, f2, cv))
    q0h = q0.cpu().numpy()
    oracle.set_eigensolver_scheme(es_scheme)
    try:
        o = oracle.solve_chain_batch(off, f1, f2, cv, q0h, seed=1, num_threads=cores)
        omask = o["mask"]
        same_mask = np.array([(omask[off[p]:off[p + 1]] == gmask[off[p]:off[p + 1]]).all() for p in range(P)])
        ang = _angles(gq, o["q"])
        ang_es = _angles(qr.cpu().numpy(), o["es_q"])
        report = {"what": label, "eigensolver_scheme": es_scheme, "pairs": P, "corr_min_max": [int(np.diff(off).min()), int(np.diff(off).max())],
                  "oracle_threads": cores, "inlier_masks_identical": int(same_mask.sum()),
                  "ransac_iteration_counts_identical": int((its.cpu().numpy() == o["ransac_iterations"]).sum()),
                  "ls_iteration_counts_identical": int((res.iterations.cpu().numpy() == o["ls_iterations"]).sum()),
                  "median_rot_diff_rad": float(np.median(ang)), "p99_rot_diff_rad": float(np.percentile(ang, 99)),
                  "max_rot_diff_rad": float(ang.max()), "pairs_over_1e-6_rad": int((ang > TOL).sum()),
                  "eigensolver_stage_pairs_over_1e-8_rad_with_identical_masks": int((same_mask & (ang_es > 1e-8)).sum()),
                  "explained": []}
        gqr, gtr, gqw, gtw = (x.cpu().numpy() for x in (qr, tr, qw, tw))
        git = res.iterations.cpu().numpy()
        L = oracle.lib()

        def stall(p):
            """scheme 2: the checker's own eigensolver call on this pair's inliers ended in a stall (1) or at maxfev (5), or
            the device's eigensolver-stage rotation is not a stationary point of the function it minimises"""
            sl = slice(off[p], off[p + 1])
            _, _, m, _ = oracle.ransac_eigensolver(f1[sl], f2[sl], oracle.rot_from_quat(q0h[p]), seed=1, pair_id=int(p))
            if L.pnec_oracle_es_last_info() in (1, 5):
                return True
            G = oracle.sums36(f1[sl][m], f2[sl][m])
            grad = oracle.es_value_grad_sums(G, oracle.rot_to_cayley(oracle.rot_from_quat(gqr[p])), reduced=True)[1]
            return bool(np.abs(grad).max() > 1e-9 * (off[p + 1] - off[p]))

        if es_scheme == 0:
            assert same_mask.mean() >= 0.9999, report
            assert np.percentile(ang, 99) <= 1e-10, report
            over = np.flatnonzero(ang > TOL)
            assert len(over) <= max(3, P // 5000), report
        else:
            assert same_mask.mean() >= 0.998, report
            assert np.percentile(ang[same_mask], 99) <= 1e-8, report
            over = np.flatnonzero(same_mask & (ang > TOL))
            assert len(over) <= 0.002 * P, report
            # the eigensolver stage by itself (what the odometry's options return): beyond 1e-8 only where the iteration stalls
            far = np.flatnonzero(same_mask & (ang_es > 1e-8))
            assert len(far) <= 0.015 * P, report
            unexplained = [int(p) for p in far if not stall(int(p))]
            report["eigensolver_stage_unexplained"] = unexplained
            assert not unexplained, report
        ang_w = _angles(gqw, o["w_q"])
        failures = []
        for p in over:
            sl = slice(off[p], off[p + 1])
            m = gmask[sl]
            entry = {"pair": int(p), "rot_diff_rad": float(ang[p]), "eigensolver_stage_diff_rad": float(ang_es[p]),
                     "weighted_stage_diff_rad": float(ang_w[p]),
                     "ls_iterations_device_oracle": [int(git[p]), int(o["ls_iterations"][p])]}
            if es_scheme == 0:
                if not same_mask[p]:
                    # the RANSAC stage chose another hypothesis: the oracle's weighted stage + refinement from the
                    # DEVICE's inliers and eigensolver pose must land on the device's final pose
                    Rw, tww = oracle.weighted_eigensolver(f1[sl][m], f2[sl][m], cv[sl][m], oracle.rot_from_quat(gqr[p]), gtr[p])
                    s = oracle.solve(oracle.MODE_TARGET, f1[sl][m], f2[sl][m], cv[sl][m], None, 1e-13, oracle.quat_from_rot(Rw),
                                     tww, oracle.default_options())
                    entry["kind"] = "ransac hypothesis bifurcation"
                else:
                    # identical inliers, refinement stopped elsewhere: the oracle's refinement from the DEVICE's
                    # weighted-stage pose must reproduce the device's iteration count (+-1) and pose
                    s = oracle.solve(oracle.MODE_TARGET, f1[sl][m], f2[sl][m], cv[sl][m], None, 1e-13, gqw[p], gtw[p],
                                     oracle.default_options())
                    entry["kind"] = "refinement stopping rule on an ill-conditioned pair"
                    if abs(int(s.iterations) - int(git[p])) > 1:
                        failures.append((int(p), "refinement iteration count", int(s.iterations), int(git[p])))
                d = float(np.radians(oracle.rotational_difference_deg(s.R, oracle.rot_from_quat(gq[p]))))
            else:
                # scheme 2, identical inliers: stage by stage from the DEVICE's own intermediates.  The refinement from the
                # device's weighted-stage pose must land on the device's pose; where the weighted stage (or the eigensolver
                # stage in front of it) is already apart, that stage's LM-on-the-gradient must have stalled.
                s = oracle.solve(oracle.MODE_TARGET, f1[sl][m], f2[sl][m], cv[sl][m], None, 1e-13, gqw[p], gtw[p],
                                 oracle.default_options())
                d = float(np.radians(oracle.rotational_difference_deg(s.R, oracle.rot_from_quat(gq[p]))))
                if abs(int(s.iterations) - int(git[p])) > 1:
                    failures.append((int(p), "refinement iteration count", int(s.iterations), int(git[p])))
                Rw, tww = oracle.weighted_eigensolver(f1[sl][m], f2[sl][m], cv[sl][m], oracle.rot_from_quat(gqr[p]), gtr[p])
                info_w = int(L.pnec_oracle_es_last_info())
                a_w = float(np.radians(oracle.rotational_difference_deg(Rw, oracle.rot_from_quat(gqw[p]))))
                entry["weighted_stage_diff_rad_from_the_devices_eigensolver_pose"] = a_w
                entry["checker_minpack_info_of_the_weighted_stages_last_minimisation"] = info_w
                if ang_es[p] > 1e-8:
                    entry["kind"] = "eigensolver stage: the iteration stalled (MINPACK info 1 / 5 on the checker's side, or a non-stationary end)"
                elif a_w > 1e-8:
                    entry["kind"] = "weighted stage: its eigenvalue minimisation stalled"
                    if info_w not in (1, 5):
                        failures.append((int(p), "weighted stage apart without a stall on the checker's side", a_w, info_w))
                else:
                    entry["kind"] = "refinement stopping rule on an ill-conditioned pair"
            entry["rot_diff_rad_from_the_devices_own_intermediate"] = d
            report["explained"].append(entry)
            if d > TOL:
                failures.append((int(p), "the last stage from the device's own intermediate does not reproduce the device", d))
        report["unexplained"] = failures
        assert not failures, report
    finally:
        oracle.set_eigensolver_scheme(0)
    # every pair that is NOT beyond the tolerance is, well, within it (the north star's bar)
    if es_scheme == 0:
        assert (np.delete(ang, over) <= TOL).all()
    else:
        assert (np.delete(ang, np.concatenate([over, np.flatnonzero(~same_mask)])) <= TOL).all()
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, f"chain_parity_at_scale_{label}_scheme{es_scheme}.json"), "w") as f:
            json.dump(report, f)
    except OSError:
        pass
    print(json.dumps(report))
    return report


@pytest.mark.parametrize("es_scheme", [0, 2])
def test_one_call_chain_matches_the_oracle_chain_over_20000_pairs(oracle, es_scheme):
    """20 000 x 512, 10 % gross mismatches, the reference's default Options -- under the C ABI's default eigensolver scheme
    (0) and under the facade's (2, the restatement believed to be what opengv runs)."""
    cores = oracle.usable_threads()                       # (the cgroup quota, not the host's CPU count)
    P, N = (20_000 if cores >= 32 else 4_000), 512      # ~11 ms of oracle per pair and thread
    dev = torch.device("cuda:0")
    g = sim.generate(P, N, seed=1, device=dev)
    bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) < 0.10
    rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    g.bvs2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
    off = np.arange(P + 1, dtype=np.int64) * N
    _check_chain_against_oracle(oracle, es_scheme, off, g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3),
                                g.init_q, g.init_t, "sim20k")


def test_kitti_like_ragged_chain_under_the_facades_default_scheme(oracle):
    """BASELINE configs 3 / 5's stand-in (all KITTI 00-10 frame pairs, ragged, 10 % gross mismatches; a quarter of them on a
    small box) through the one-call chain under eigensolver scheme 2 -- what a user of the C++ facade gets -- against the
    checker's scheme-2 chain."""
    from pnec_amd import tracks as tk
    cores = oracle.usable_threads()
    P_all = int(len(tk.kitti_all_sizes()))
    P = P_all if cores >= 32 else P_all // 4
    tr = tk.kitti_all_shard(0, P, device=torch.device("cuda:0"), outlier_frac=0.10)
    _check_chain_against_oracle(oracle, 2, np.asarray(tr.offsets, dtype=np.int64), tr.bvs1, tr.bvs2, tr.covs,
                                tr.init_q.contiguous(), tr.init_t.contiguous(), "kitti_all")


def test_two_pairs_per_wavefront_ransac_is_bitwise_the_one_pair_form():
    """From 4 096 pairs up the RANSAC stage runs two pairs per wavefront with the hypotheses of both in one queue
    (ransac2_eigensolver_kernel: a quad that has finished a minimisation takes the next hypothesis, whichever pair it
    belongs to; the queue ordered by a key of the minimisations' starts, likely-long ones first; both slots' samples drawn
    in one pass; the launch's last pairs on wavefronts of their own); smaller batches keep one wavefront per pair.  A
    hypothesis' arithmetic does not depend on which quad minimises it or when, so the whole chain over 6 001 ragged pairs (odd: the last wavefront holds one pair; sizes
    from below the sample size to beyond one wavefront's 512; gross mismatches) in ONE call must equal, bit for bit,
    the same pairs solved as three batches of <= 2 048 (one-pair form) whose RANSAC draws are offset to the pairs'
    global indices (`first_pair_id`)."""
    dev = torch.device("cuda:0")
    P = 6001
    rng = np.random.default_rng(77)
    counts = rng.integers(60, 640, size=P).astype(np.int64)
    counts[:8] = [5, 9, 10, 11, 64, 512, 513, 639]          # below / at the 10-point sample, slot boundaries
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    nmax = int(counts.max())
    parts = []
    for c0 in range(0, P, 1000):                              # generated in chunks: [m, nmax] boxes, then ragged
        m = min(1000, P - c0)
        g = sim.generate(m, nmax, seed=900 + c0, device=dev)
        keep = torch.arange(nmax, device=dev)[None, :] < torch.as_tensor(counts[c0:c0 + m], device=dev)[:, None]
        bad = torch.rand(m, nmax, device=dev, generator=torch.Generator(device=dev).manual_seed(c0)) < 0.15
        rnd = torch.randn(m, nmax, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c0 + 1))
        b2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
        parts.append((g.bvs1[keep], b2[keep], g.covs2[keep], g.init_q, g.init_t))
    f1, f2, cv, q0, t0 = (torch.cat([p[i] for p in parts]) for i in range(5))
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, cv)
        q, t, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
    torch.cuda.synchronize()
    for a in range(0, P, 2048):
        z = min(P, a + 2048)
        sl = slice(int(off[a]), int(off[z]))
        with Batch(capi.MODE_TARGET, off[a:z + 1] - off[a]) as b:
            b.fill(f1[sl], f2[sl], cv[sl])
            qs, ts, ms, cs = b.solve_pipeline(q0[a:z].contiguous(), t0[a:z].contiguous(), want_inliers=True,
                                              options=capi.default_pipeline_options(first_pair_id=a))
        assert torch.equal(ms, mask[sl]) and torch.equal(cs, cnt[a:z]), (a, z)
        assert torch.equal(qs, q[a:z]) and torch.equal(ts, t[a:z]), (a, z)
    assert int(cnt[:3].max()) <= 10 and bool((cnt[8:] > 30).all())          # tiny pairs: no sampling / all inliers


@pytest.mark.gpu
def test_launch_order_hint_changes_the_dispatch_order_and_no_bit(oracle):
    """pnec_hip_problem_launch_order_hint: the RANSAC stage of the next call dispatches the pairs that took more than one
    round of hypotheses in the last one first.  Scheduling only: poses, masks and counts of the hinted call are those of
    the unhinted one, bit for bit -- on a batch large enough for the two-pair form that honours the order, with 30 % gross
    outliers so that a good part of the pairs goes beyond one round (the order is far from the identity), through both
    entries that run RANSAC, and again after the batch's contents (not its shape) were replaced (a stale hint)."""
    import torch
    from pnec_amd import Batch, capi
    from pnec_amd import simulation as sim
    dev = torch.device("cuda:0")
    P, N = 4600, 160
    def make(seed):
        g = sim.generate(P, N, seed=seed, device=dev)
        bad = torch.rand(P, N, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)) < 0.3
        rnd = torch.randn(P, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(seed + 1))
        b2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
        return g, b2
    g, b2 = make(31)
    batch = Batch.uniform(capi.MODE_TARGET, P, N)
    batch.fill(g.bvs1.reshape(-1, 3), b2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    q_ref, t_ref, m_ref, c_ref = batch.solve_pipeline(g.init_q, g.init_t, want_inliers=True)
    qr_ref, tr_ref, mr_ref, cr_ref, it_ref = batch.ransac_eigensolver(g.init_q, seed=1)
    assert int((it_ref > 16).sum()) > P // 20                      # the order will not be the identity
    batch.launch_order_hint(True)
    for _ in range(2):                                             # first call: no hint yet; second: hinted
        q, t, m, c = batch.solve_pipeline(g.init_q, g.init_t, want_inliers=True)
        assert torch.equal(q, q_ref) and torch.equal(t, t_ref) and torch.equal(m, m_ref) and torch.equal(c, c_ref)
    for _ in range(2):
        qr, tr, mr, cr, it = batch.ransac_eigensolver(g.init_q, seed=1)
        assert torch.equal(qr, qr_ref) and torch.equal(mr, mr_ref) and torch.equal(it, it_ref)
    # new contents, same shape: the hint is stale, the results are the new data's
    g2, b22 = make(77)
    batch.fill(g2.bvs1.reshape(-1, 3), b22.reshape(-1, 3), g2.covs2.reshape(-1, 3, 3))
    q_h, t_h, m_h, c_h = batch.solve_pipeline(g2.init_q, g2.init_t, want_inliers=True)
    batch.launch_order_hint(False)
    q_n, t_n, m_n, c_n = batch.solve_pipeline(g2.init_q, g2.init_t, want_inliers=True)
    assert torch.equal(q_h, q_n) and torch.equal(t_h, t_n) and torch.equal(m_h, m_n) and torch.equal(c_h, c_n)
    batch.close()


@pytest.mark.gpu
def test_offsets_and_launch_order_beyond_one_scan_segment():
    """The two one-workgroup index kernels walk the pairs in segments of 32 x 1024 (offsets_scan_kernel: AoS offsets of an
    InlierExtraction target; ransac_order_kernel: the hinted launch order): batches of more pairs than one segment -- the
    offsets are the running sum of the kept counts, and the hinted RANSAC call returns the unhinted one's bits (an order
    that is not a permutation loses or repeats pairs)."""
    import torch
    from pnec_amd import Batch, capi
    from pnec_amd import simulation as sim
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    P = 70001
    sizes = rng.integers(3, 10, size=P)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    M = int(off[-1])
    f = rng.standard_normal((2, M, 3))
    f /= np.linalg.norm(f, axis=2, keepdims=True)
    mask = (rng.random(M) < 0.6).astype(np.uint8)
    with Batch(capi.MODE_NEC, off) as b:
        b.fill(f[0], f[1], None)
        sel = b.select(mask)
        kept = np.add.reduceat(mask.astype(np.int64), off[:-1])
        assert np.array_equal(sel.offsets, np.concatenate([[0], np.cumsum(kept)]))
        sel.close()
    P2, N2 = 40000, 48
    g = sim.generate(P2, N2, seed=3, device=dev)
    bad = torch.rand(P2, N2, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) < 0.3
    rnd = torch.randn(P2, N2, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    b2 = torch.where(bad[..., None], rnd / rnd.norm(dim=-1, keepdim=True), g.bvs2)
    batch = Batch.uniform(capi.MODE_TARGET, P2, N2)
    batch.fill(g.bvs1.reshape(-1, 3), b2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    ref = batch.ransac_eigensolver(g.init_q, seed=1)
    assert int((ref[4] > 16).sum()) > P2 // 20
    batch.launch_order_hint(True)
    for _ in range(2):
        out = batch.ransac_eigensolver(g.init_q, seed=1)
        for x, y in zip(out, ref):
            assert torch.equal(x, y)
    batch.close()
