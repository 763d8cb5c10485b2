"""The reference's own surfaces for this path: the pybind module `pypnec` (pyceres / pyceresnec,
python/pypnec.cpp:50-82) and the C++ classes behind it, served by the HIP solver."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from pnec_amd import simulation as sim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pnec_amd"))


@pytest.fixture(autouse=True)
def _facade_default_scheme(request):
    """The facade defaults to eigensolver scheme 2 (pnec_host.h Options::eigensolver_scheme_: the restatement believed to be
    what opengv runs); the checker these tests compare with must run the same one."""
    if "oracle" not in request.fixturenames:
        yield
        return
    po = request.getfixturevalue("oracle")
    po.set_eigensolver_scheme(2)
    yield
    po.set_eigensolver_scheme(0)


def _pose4(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def test_pypnec_module_surface():
    import pypnec
    assert pypnec.add(2, 3) == 5
    for name in ("pyceres", "pyceresnec", "ceres_solver_batch", "solve_batch"):
        assert callable(getattr(pypnec, name))
    with pytest.raises(ValueError):   # argument validation happens before any device work
        pypnec.pyceresnec(np.zeros((4, 2)), np.zeros((4, 3)), np.eye(4))


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_pypnec_fails_loudly_without_gpu():
    import pypnec
    g = sim.generate(1, 16, seed=1)
    with pytest.raises(RuntimeError):
        pypnec.pyceresnec(list(g.bvs1[0].numpy()), list(g.bvs2[0].numpy()),
                          _pose4(g.init_R[0].numpy(), g.init_t[0].numpy()))


@pytest.mark.gpu
def test_pyceres_matches_oracle_symmetric(oracle):
    """pyceres = PNECCeres::Optimize(bvs1, bvs2, covs_1, covs_2, reg): call it the way the reference's
    Python users do (lists of arrays) and compare with the reference-faithful CPU oracle."""
    import pypnec
    g = sim.generate(3, 200, seed=77)
    for p in range(3):
        f1, f2, S2 = g.bvs1[p].numpy(), g.bvs2[p].numpy(), g.covs2[p].numpy()
        S1 = np.roll(S2, 2, axis=0) * 0.6
        init = _pose4(g.init_R[p].numpy(), g.init_t[p].numpy())
        T = pypnec.pyceres(list(f1), list(f2), list(S1), list(S2), init, 1e-13)
        s = oracle.solve(oracle.MODE_SYM, f1, f2, S2, S1, 1e-13, oracle.quat_from_rot(init[:3, :3]),
                         init[:3, 3], oracle.default_options())
        assert T.shape == (4, 4) and np.allclose(T[3], [0, 0, 0, 1])
        assert math.radians(oracle.rotational_difference_deg(T[:3, :3], s.R)) <= 1e-6
        assert abs(T[:3, 3] @ s.t) > 1 - 1e-10


@pytest.mark.gpu
def test_pyceresnec_and_batch_match_oracle(oracle):
    import pypnec
    g = sim.generate(4, 150, seed=78)
    poses = [_pose4(g.init_R[p].numpy(), g.init_t[p].numpy()) for p in range(4)]
    T = pypnec.pyceresnec(g.bvs1[0].numpy(), g.bvs2[0].numpy(), poses[0])
    s = oracle.solve(oracle.MODE_NEC, g.bvs1[0].numpy(), g.bvs2[0].numpy(), None, None, 0.0,
                     oracle.quat_from_rot(poses[0][:3, :3]), poses[0][:3, 3], oracle.default_options())
    assert math.radians(oracle.rotational_difference_deg(T[:3, :3], s.R)) <= 1e-6
    # ragged batch through PNEC::CeresSolverBatch
    sizes = [150, 100, 64, 33]
    out = pypnec.ceres_solver_batch([g.bvs1[p, :n].numpy() for p, n in enumerate(sizes)],
                                    [g.bvs2[p, :n].numpy() for p, n in enumerate(sizes)],
                                    [g.covs2[p, :n].numpy() for p, n in enumerate(sizes)], poses, 1e-13)
    for p, n in enumerate(sizes):
        s = oracle.solve(oracle.MODE_TARGET, g.bvs1[p, :n].numpy(), g.bvs2[p, :n].numpy(),
                         g.covs2[p, :n].numpy(), None, 1e-13, oracle.quat_from_rot(poses[p][:3, :3]),
                         poses[p][:3, 3], oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(out[p][:3, :3], s.R)) <= 1e-6


@pytest.mark.gpu
def test_cpp_facade_demo_runs_run_simulation_call_pattern(tmp_path, oracle):
    """BASELINE config 1 through the C++ facade with no Python in the product path: the demo executable makes
    run_simulation's call (run_simulation.cc:74-86: PNEC::Solve(bvs1, bvs2, covs, init, inliers) with the
    reference's default Options) on ONE pair of 100 isotropic-covariance correspondences, dumps its inputs and
    its result, and the oracle's chain on those very inputs must give the same inliers and the same pose."""
    exe = os.path.join(ROOT, "pnec_amd", "pnec_host_demo")
    dump = tmp_path / "pair.txt"
    r = subprocess.run([exe, "100", "dump", str(dump)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = open(dump).read().splitlines()
    n = int(rows[0])
    assert n == 100
    a = np.array([[float(v) for v in row.split()] for row in rows[1:1 + n]])
    b1, b2 = np.ascontiguousarray(a[:, 0:3]), np.ascontiguousarray(a[:, 3:6])
    cv = np.ascontiguousarray(a[:, 6:15].reshape(n, 3, 3))
    assert np.allclose(cv, cv[0, 0, 0] * np.eye(3))                      # isotropic, the same for every point
    init = np.array([float(v) for v in rows[1 + n].split()])
    sol = np.array([float(v) for v in rows[2 + n].split()])
    inl = [int(v) for v in rows[3 + n].split()]
    assert inl[0] == len(inl) - 1
    R0 = oracle.rot_from_quat(init[:4])
    Ro, to, mo, _ = oracle.ransac_eigensolver(b1, b2, R0, seed=1, pair_id=0)
    assert inl[1:] == list(np.flatnonzero(mo))
    Rw, tw = oracle.weighted_eigensolver(b1[mo], b2[mo], cv[mo], Ro, to, 1e-13, 10)
    s = oracle.solve(oracle.MODE_TARGET, b1[mo], b2[mo], cv[mo], None, 1e-13, oracle.quat_from_rot(Rw), tw,
                     oracle.default_options())
    R = oracle.rot_from_quat(sol[:4])
    assert math.radians(oracle.rotational_difference_deg(R, s.R)) <= 1e-6          # the north star's tolerance
    assert abs(sol[4:7] @ s.t) > 1 - 1e-10
    # the line the demo prints carries the facade's metric helpers: same numbers from the oracle's
    f = dict(kv.split("=") for kv in r.stdout.split())
    assert int(f["inliers"]) == inl[0]
    assert float(f["cost"]) == pytest.approx(oracle.cost_function(b1, b2, cv, R, sol[4:7]), rel=1e-5, abs=1e-6)
    assert float(f["rot_err_deg"]) < float(f["rot_err_init_deg"])


@pytest.mark.gpu
def test_cpp_facade_demo_writes_timing_txt_like_the_odometry(tmp_path):
    """pnec_vo.cc:220-261,273-276: a FrameTiming per frame filled by the timed PNEC::Solve overload, collected in
    a Timing, streamed into timing.txt -- header line, then one row of integral milliseconds per frame whose
    OPTIMIZATION / TOTAL columns are the sums timing.cc:40-47 define."""
    from pnec_amd import io_formats as io
    exe = os.path.join(ROOT, "pnec_amd", "pnec_host_demo")
    path = tmp_path / "timing.txt"
    r = subprocess.run([exe, "200", "timing", "6", str(path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(path).read()
    assert text.splitlines()[0] == io.TIMING_HEADER and text.endswith("\n")
    a = io.read_timing_file(path)                       # checks the derived columns
    assert a.shape == (6, 9) and a[:, 0].tolist() == [1, 2, 3, 4, 5, 6]
    assert (a[:, 1:3] == 0).all()                       # no frame loading / feature creation on this path
    assert (a >= 0).all() and (a[:, 5] == a[:, 4] // 10).all()          # avg-it-es = it-es / weighted_iterations
    assert all(len(line.split(" ")) == 9 and all(c.isdigit() for c in line.split(" ")) for line in text.splitlines()[1:])


def test_common_helpers_of_the_facade_match_oracle(oracle):
    """pnec::common::ComposeM (loop from i = 1) and TranslationFromM in the C++ facade"""
    import pypnec
    g = sim.generate(1, 60, seed=3)
    f1, f2, R = g.bvs1[0].numpy(), g.bvs2[0].numpy(), g.R_gt[0].numpy()
    M = pypnec.compose_m(f1, f2, R)
    np.testing.assert_allclose(M, oracle.compose_m(f1, f2, R, skip_first=True), atol=1e-14)
    np.testing.assert_allclose(pypnec.translation_from_m(M), oracle.translation_from_m(M), atol=1e-13)


@pytest.mark.gpu
def test_solve_batch_runs_the_default_pipeline_for_a_ragged_batch(oracle):
    """PNEC::SolveBatch = PNEC::Solve (pnec.cc:77-124, reference-default Options) for many pairs, one
    launch per stage: RANSAC inliers, weighted eigensolver + SCF, refinement -- against the oracle's
    chain with the same counter-based draws (pair p of the batch draws as pair_id = p)."""
    import pypnec
    sizes = [200, 512, 90]
    g = sim.generate(3, 512, seed=321)
    rng = np.random.default_rng(5)
    b1 = [g.bvs1[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    b2 = [g.bvs2[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    cv = [g.covs2[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    for p, n in enumerate(sizes):                      # 15 % gross outliers
        bad = rng.choice(n, n * 15 // 100, replace=False)
        v = rng.normal(size=(len(bad), 3))
        b2[p][bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    poses = [_pose4(g.init_R[p].numpy(), g.init_t[p].numpy()) for p in range(3)]
    out, inliers = pypnec.solve_batch(b1, b2, cv, poses)
    for p, n in enumerate(sizes):
        Ro, to, mo, _ = oracle.ransac_eigensolver(b1[p], b2[p], poses[p][:3, :3], seed=1, pair_id=p)
        assert list(inliers[p]) == list(np.flatnonzero(mo))
        Rw, tw = oracle.weighted_eigensolver(b1[p][mo], b2[p][mo], cv[p][mo], Ro, to, 1e-13, 10)
        s = oracle.solve(oracle.MODE_TARGET, b1[p][mo], b2[p][mo], cv[p][mo], None, 1e-13,
                         oracle.quat_from_rot(Rw), tw, oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(out[p][:3, :3], s.R)) <= 1e-6
        assert math.radians(oracle.rotational_difference_deg(out[p][:3, :3], g.R_gt[p].numpy())) < 0.01
    # the flags: NEC refinement without RANSAC, and eigensolver only
    out2, inl2 = pypnec.solve_batch(b1, b2, cv, poses, use_ransac=False, use_nec=True)
    assert all(len(i) == 0 for i in inl2)
    for p in range(3):
        Ro, to = oracle.nec_eigensolver(b1[p], b2[p], poses[p][:3, :3])
        s = oracle.solve(oracle.MODE_NEC, b1[p], b2[p], None, None, 0.0, oracle.quat_from_rot(Ro), to,
                         oracle.default_options())
        assert math.radians(oracle.rotational_difference_deg(out2[p][:3, :3], s.R)) <= 1e-6


@pytest.mark.gpu
def test_solve_batch_survives_degenerate_pairs():
    """empty / tiny / NaN / all-identical / zero-covariance / outlier-dominated pairs in one ragged
    batch, under every Options branch of PNEC::Solve: no error, one pose per pair, and the clean
    pairs are unaffected by their neighbours"""
    import pypnec
    rng = np.random.default_rng(123)
    B = 40
    sizes = [int(x) for x in rng.choice([0, 1, 5, 9, 10, 11, 30, 64, 65, 128, 129, 300, 512, 513, 700], size=B)]
    g = sim.generate(B, 700, seed=500)
    b1 = [g.bvs1[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    b2 = [g.bvs2[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    cv = [g.covs2[p, :n].numpy().copy() for p, n in enumerate(sizes)]
    clean = []
    for p, n in enumerate(sizes):
        kind = int(rng.integers(0, 5)) if n > 0 else 4
        if kind == 0:
            bad = rng.choice(n, max(1, n * 7 // 10), replace=False)
            v = rng.normal(size=(len(bad), 3))
            b2[p][bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
        elif kind == 1:
            b2[p][rng.integers(0, n)] = np.nan
        elif kind == 2:
            b1[p][:] = b1[p][0]
            b2[p][:] = b2[p][0]
        elif kind == 3:
            cv[p][:] = 0.0
        clean.append(kind == 4 and n >= 64)
    assert sum(clean) >= 3
    poses = [_pose4(g.init_R[p].numpy(), g.init_t[p].numpy()) for p in range(B)]
    for kw in (dict(), dict(use_ransac=False), dict(use_nec=True), dict(weighted_iterations=1), dict(use_ceres=False)):
        out, inl = pypnec.solve_batch(b1, b2, cv, poses, **kw)
        assert len(out) == B and len(inl) == B
        for p in range(B):
            if clean[p]:
                assert np.isfinite(out[p]).all()
                R = out[p][:3, :3]
                assert np.allclose(R @ R.T, np.eye(3), atol=1e-9)
                err = np.degrees(np.arccos(np.clip((np.trace(R.T @ g.R_gt[p].numpy()) - 1) / 2, -1, 1)))
                assert err < 1.0, (kw, p, sizes[p], err)


def test_metric_helpers_of_the_facade_follow_the_reference_semantics(oracle):
    """pnec::common::RotationalDifference / TranslationalDifference of the C++ facade against an
    independent numpy statement of common.cc:210-235: |log(R1' R2)| in DEGREES; the angle between the
    translations in degrees, the smaller of t2 / -t2 when both_directions, 90 when translation_1 is
    (nearly) zero -- and, because the reference tests translation_1.norm() twice (quirk C8), NaN rather
    than 90 when only translation_2 is zero."""
    import pypnec
    rng = np.random.default_rng(4)

    def rot(axis, ang):
        a = np.asarray(axis, float) / np.linalg.norm(axis)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * K @ K

    for ang in (0.0, 1e-9, 1e-4, 0.3, 1.5, 3.0, math.pi - 1e-6):
        for _ in range(4):
            R1 = rot(rng.normal(size=3), rng.uniform(0, 3))
            R2 = R1 @ rot(rng.normal(size=3), ang)
            got = pypnec.rotational_difference(R1, R2)
            assert got == pytest.approx(math.degrees(ang), abs=1e-9 if ang > 1e-6 else 1e-12)
            assert got == pytest.approx(oracle.rotational_difference_deg(R1, R2), abs=1e-9)
            assert got >= 0.0
    for _ in range(20):
        t1, t2 = rng.normal(size=3) * rng.uniform(0.1, 5), rng.normal(size=3) * rng.uniform(0.1, 5)
        c = t1 @ t2 / (np.linalg.norm(t1) * np.linalg.norm(t2))
        one = math.degrees(math.acos(c))
        assert pypnec.translational_difference(t1, t2, False) == pytest.approx(one, abs=1e-10)
        assert pypnec.translational_difference(t1, t2, True) == pytest.approx(min(one, 180.0 - one), abs=1e-10)
        assert pypnec.translational_difference(t1, t2) == pytest.approx(min(one, 180.0 - one), abs=1e-10)  # default
    z = np.zeros(3)
    e = np.array([0.0, 0.0, 1.0])
    assert pypnec.translational_difference(z, e) == 90.0            # M_PI / 2 branch
    assert pypnec.translational_difference(1e-11 * e, e) == 90.0
    assert math.isnan(pypnec.translational_difference(e, z))        # C8: the second norm is never tested
    assert pypnec.translational_difference(e, -e, True) == pytest.approx(0.0, abs=1e-6)
    assert pypnec.translational_difference(e, -e, False) == pytest.approx(180.0, abs=1e-6)


@pytest.mark.gpu
def test_ransac_chained_starts_through_the_facade(oracle):
    """Options::ransac_chained_starts_ (PNEC_HIP_RANSAC_CHAINED_STARTS [EXT]): the untimed overloads pass it in the chain's
    options, the timed ones set it on the batch for the stage call -- same pose, same inliers; the inliers are the
    checker's with its same switch."""
    import pypnec
    g = sim.generate(1, 300, seed=411)
    rng = np.random.default_rng(6)
    b1, b2, cv = g.bvs1[0].numpy().copy(), g.bvs2[0].numpy().copy(), g.covs2[0].numpy().copy()
    bad = rng.choice(300, 75, replace=False)
    v = rng.normal(size=(75, 3))
    b2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    init = _pose4(g.init_R[0].numpy(), g.init_t[0].numpy())
    T1, inl1, _ = pypnec.solve(b1, b2, cv, init, overload=1, ransac_chained_starts=True)
    T3, inl3, _ = pypnec.solve(b1, b2, cv, init, overload=3, ransac_chained_starts=True)
    T0, inl0, _ = pypnec.solve(b1, b2, cv, init, overload=1)
    np.testing.assert_array_equal(T1, T3)
    assert inl1 == inl3 and 150 < len(inl1) <= 225
    oracle.set_eigensolver_scheme(2)
    oracle.set_ransac_chained_starts(True)
    try:
        _, _, mo, its_on = oracle.ransac_eigensolver(b1, b2, init[:3, :3], seed=1, pair_id=0)
        oracle.set_ransac_chained_starts(False)
        _, _, mf, its_off = oracle.ransac_eigensolver(b1, b2, init[:3, :3], seed=1, pair_id=0)
    finally:
        oracle.set_ransac_chained_starts(False)
        oracle.set_eigensolver_scheme(0)
    assert inl1 == list(np.flatnonzero(mo)) and inl0 == list(np.flatnonzero(mf))
    # (that the switch moves hypothesis counts and masks: tests/test_opengv_schemes.py on 600 pairs; this pair's are equal)


@pytest.mark.gpu
def test_all_four_solve_overloads_agree_and_fill_the_timing_like_the_reference(oracle):
    """PNEC::Solve has four overloads (pnec.cc:69-75, :77-124, :126-134, :135-208): same pose from all,
    the inlier list from the two that take one, and a FrameTiming whose fields are written exactly where
    the reference writes them (nec_es_ always; it_es_ in the PNEC branch; avg_it_es_ only when the
    weighted stage runs; ceres_ always)."""
    import pypnec
    g = sim.generate(1, 300, seed=411)
    rng = np.random.default_rng(6)
    b1, b2, cv = g.bvs1[0].numpy().copy(), g.bvs2[0].numpy().copy(), g.covs2[0].numpy().copy()
    bad = rng.choice(300, 45, replace=False)
    v = rng.normal(size=(45, 3))
    b2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    init = _pose4(g.init_R[0].numpy(), g.init_t[0].numpy())
    outs = [pypnec.solve(b1, b2, cv, init, overload=k) for k in range(4)]
    for T, inl, tim in outs[1:]:
        np.testing.assert_array_equal(T, outs[0][0])              # bit-identical poses
    assert outs[0][1] is None and outs[0][2] is None and outs[2][1] is None
    assert outs[1][1] == outs[3][1] and 150 < len(outs[1][1]) <= 300 - 40
    assert len(set(outs[1][1]) & set(bad.tolist())) <= 2          # gross outliers are (all but) gone
    # against the oracle chain (pair_id 0, seed 1 -- what a single Solve draws)
    Ro, to, mo, _ = oracle.ransac_eigensolver(b1, b2, init[:3, :3], seed=1, pair_id=0)
    assert outs[1][1] == list(np.flatnonzero(mo))
    Rw, tw = oracle.weighted_eigensolver(b1[mo], b2[mo], cv[mo], Ro, to, 1e-13, 10)
    s = oracle.solve(oracle.MODE_TARGET, b1[mo], b2[mo], cv[mo], None, 1e-13, oracle.quat_from_rot(Rw), tw,
                     oracle.default_options())
    assert math.radians(oracle.rotational_difference_deg(outs[0][0][:3, :3], s.R)) <= 1e-6
    for T, inl, tim in (outs[2], outs[3]):
        assert tim["id"] == 7 and tim["header"] == "ID FrameLoading FeatureCreation NEC-ES IT-ES AVG-IT-ES CERES OPTIMIZATION TOTAL"
        assert min(tim["nec_es"], tim["it_es"], tim["avg_it_es"], tim["ceres"]) >= 0     # all written (ms)
        assert tim["avg_it_es"] == tim["it_es"] // 10
        assert tim["optimization"] == tim["nec_es"] + tim["it_es"] + tim["ceres"] and tim["total"] == tim["optimization"]
        # the microsecond twins (round 4): every stage took SOME time, the millisecond fields are their truncations,
        # and the row prints them with the id in front
        for k in ("nec_es", "it_es", "ceres"):
            assert 1.0 < tim[k + "_us"] < 5e5 and tim[k] == int(tim[k + "_us"] / 1000.0), (k, tim)
        assert tim["avg_it_es_us"] == pytest.approx(tim["it_es_us"] / 10) and tim["optimization_us"] == pytest.approx(
            tim["nec_es_us"] + tim["it_es_us"] + tim["ceres_us"])
        assert tim["row_us"].split()[0] == "7" and len(tim["row_us"].split()) == len(tim["header_us"].split()) == 6
    # which fields each Options branch writes (-1 = left as the caller had it)
    _, _, tim = pypnec.solve(b1, b2, cv, init, overload=3, use_nec=True)
    assert tim["nec_es"] >= 0 and tim["ceres"] >= 0 and tim["it_es"] == -1 and tim["avg_it_es"] == -1
    _, _, tim = pypnec.solve(b1, b2, cv, init, overload=3, use_nec=True, use_ceres=False)
    assert tim["ceres"] == 0 and tim["it_es"] == -1
    _, _, tim = pypnec.solve(b1, b2, cv, init, overload=2, weighted_iterations=1)
    assert tim["it_es"] == 0 and tim["avg_it_es"] == -1 and tim["ceres"] >= 0
    _, _, tim = pypnec.solve(b1, b2, cv, init, overload=2, weighted_iterations=0, use_ceres=False)
    assert tim["it_es"] == 0 and tim["avg_it_es"] == -1 and tim["ceres"] == 0
    # CostFunction of the facade (device) against the oracle; empty input is NaN like the reference
    T = outs[0][0]
    want = oracle.cost_function(b1, b2, cv, T[:3, :3], T[:3, 3])
    assert pypnec.cost_function(b1, b2, cv, T) == pytest.approx(want, rel=1e-10)
    assert math.isnan(pypnec.cost_function(np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3, 3)), T))
