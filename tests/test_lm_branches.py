"""Branches of the Levenberg-Marquardt restatement that ordinary data never reaches, executed on purpose -- on the CPU
oracle here, and on the device (-m gpu) against it -- and the device's verification mode that differentiates the way the
reference does (ceres::NumericDiffCostFunction<..., CENTRAL, ...>, src/optimization/pnec_ceres.cc:84-97).

[EXT, recalled] marks Ceres 2.x behaviour restated from memory (Ceres is not in the reference tree):
  * TrustRegionMinimizer::HandleInvalidStep -> LevenbergMarquardtStrategy::StepIsInvalid(): radius *= 0.5, the diagonal
    is reused, decrease_factor is NOT touched; max_num_consecutive_invalid_steps of them end the solve;
  * IterationZero() leaves step_is_successful = true, so a start point whose gradient is below the tolerance ends the
    solve after iteration 0 ("Gradient tolerance reached").
"""
import math

import numpy as np
import pytest


def _zero_gradient_pair():
    """A pair whose every residual is EXACTLY zero at the start pose, with a Jacobian that is not: f1 = f2 with small
    integer components (every product below is exact in binary64, fused or not), R = I, t = e_z (theta = phi = 0:
    sin / cos exact).  Then n_i = t . (f1 x R f2) = 0, g = J'r = 0, the LM step is 0 and its model cost change 0 -- which
    Ceres calls an INVALID step (model_cost_change > 0 is the test)."""
    f = np.array([[1, 2, 2], [2, -1, 2], [-2, 2, 1], [3, 0, 4], [0, 3, 4], [1, -2, 2], [2, 2, -1], [4, 0, 3]], dtype=np.float64)
    cov = np.tile(np.diag([1.0, 2.0, 4.0]) * 2.0 ** -10, (len(f), 1, 1))
    q0 = np.array([0.0, 0.0, 0.0, 1.0])
    t0 = np.array([0.0, 0.0, 1.0])
    return f, f.copy(), cov, q0, t0


@pytest.mark.parametrize("mode_name", ["TARGET", "NEC"])
def test_oracle_executes_the_invalid_step_branch(oracle, mode_name):
    po = oracle
    mode = getattr(po, "MODE_" + mode_name)
    f1, f2, cov, q0, t0 = _zero_gradient_pair()
    c2 = None if mode == po.MODE_NEC else cov
    for a in range(len(f1)):                           # the construction holds: every residual is exactly zero
        assert po.residual(mode, f1[a], f2[a], None if c2 is None else po.covs_to_colmajor9(c2[a:a + 1])[0], None, 1e-13,
                           0.0, 0.0, q0) == 0.0
    po.lm_diagnostics(True)
    try:
        po.lm_invalid_steps(reset=True)
        s = po.solve(mode, f1, f2, c2, None, 1e-13, q0, t0, po.default_options(check_convergence=0, max_num_iterations=50))
        n_invalid = po.lm_invalid_steps(reset=True)
    finally:
        po.lm_diagnostics(False)
    # five consecutive invalid steps (the default bound) end the solve; each was an iteration (Ceres counts them)
    assert s.status == 5 and s.iterations == 5, (s.status, s.iterations)
    assert n_invalid == 4                              # the fifth ends the solve before the radius is touched again
    assert s.cost == 0.0 and np.array_equal(s.q, q0)
    # with the convergence tests on, iteration zero's gradient (exactly 0 <= 1e-10) ends the solve first
    s = po.solve(mode, f1, f2, c2, None, 1e-13, q0, t0, po.default_options())
    assert s.status == 2 and s.iterations == 0, (s.status, s.iterations)
    # a smaller bound on consecutive invalid steps is honoured
    s = po.solve(mode, f1, f2, c2, None, 1e-13, q0, t0,
                 po.default_options(check_convergence=0, max_num_iterations=50, max_num_consecutive_invalid_steps=2))
    assert s.status == 5 and s.iterations == 2


@pytest.mark.gpu
@pytest.mark.parametrize("mode_name", ["TARGET", "NEC"])
def test_device_executes_the_invalid_step_branch_like_the_oracle(oracle, mode_name):
    from pnec_amd import Batch, capi
    po = oracle
    mode = getattr(capi, "MODE_" + mode_name)
    f1, f2, cov, q0, t0 = _zero_gradient_pair()
    c2 = None if mode == capi.MODE_NEC else cov
    n = len(f1)
    # the same pair on every launch geometry that holds it, and as a member of a batch of ordinary pairs
    for tune in (dict(), dict(corr_per_lane=8, waves_per_pair=1, lds_corr_per_lane=3 if mode == capi.MODE_TARGET else 0),
                 dict(corr_per_lane=0, waves_per_pair=8)):
        for kw, want in ((dict(check_convergence=0, max_num_iterations=50), (5, 5)), (dict(), (2, 0)),
                         (dict(check_convergence=0, max_num_iterations=50, max_num_consecutive_invalid_steps=2), (5, 2)),
                         (dict(check_convergence=0, max_num_iterations=3), (3, 3))):
            with Batch(mode, np.array([0, n, 2 * n], dtype=np.int64)) as b:
                b.fill(np.concatenate([f1, f1]), np.concatenate([f2, f2]), None if c2 is None else np.concatenate([c2, c2]))
                r = b.solve(np.stack([q0, q0]), np.stack([t0, t0]), options=capi.default_options(**kw, **tune))
            okw = {k: v for k, v in kw.items()}
            s = po.solve(mode, f1, f2, c2, None, 1e-13, q0, t0, po.default_options(**okw))
            assert (s.status, s.iterations) == want
            assert (int(r.status[0]), int(r.iterations[0])) == want, (tune, kw, r.status, r.iterations)
            assert (int(r.status[1]), int(r.iterations[1])) == want
            assert float(r.cost[0]) == 0.0 and np.array_equal(r.q[0], q0)


@pytest.mark.gpu
@pytest.mark.parametrize("mode_name", ["NEC", "TARGET", "HOST", "SYM"])
def test_numeric_jacobian_mode_follows_the_reference_path(oracle, mode_name):
    """PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL: the device differentiates as the reference does.  Against the oracle running
    the same differentiation: same iteration counts and termination codes, rotations within 1e-8 rad (two implementations
    of a difference quotient agree to ~1e-9 of the Jacobian: the quotient amplifies the residual's last bits by 1 / 2h =
    3e7); and the production kernel (closed form) stays within the north-star tolerance of both."""
    from pnec_amd import Batch, capi
    from pnec_amd import simulation as sim
    po = oracle
    mode = getattr(capi, "MODE_" + mode_name)
    B, n = 24, 100
    g = sim.generate(B, n, seed=321)
    f1, f2 = g.bvs1.reshape(-1, 3).numpy(), g.bvs2.reshape(-1, 3).numpy()
    S2 = g.covs2.reshape(-1, 3, 3).numpy()
    c2, c1 = (None, None) if mode == capi.MODE_NEC else ((S2, np.roll(S2, 1, axis=0) * 0.8) if mode == capi.MODE_SYM else (S2, None))
    offsets = np.arange(B + 1, dtype=np.int64) * n
    q0, t0 = g.init_q.numpy(), g.init_t.numpy()
    with Batch(mode, offsets) as b:
        b.fill(f1, f2, c2, c1)
        rn = b.solve(q0, t0, options=capi.default_options(flags=capi.OPT_JACOBIAN_NUMERIC_CENTRAL))
        ra = b.solve(q0, t0, options=capi.default_options())
        # the evaluation itself: zero iterations return the cost at the start, the same number either way
        r0n = b.solve(q0, t0, options=capi.default_options(max_num_iterations=0, flags=capi.OPT_JACOBIAN_NUMERIC_CENTRAL))
        r0a = b.solve(q0, t0, options=capi.default_options(max_num_iterations=0))
    np.testing.assert_allclose(r0n.cost, r0a.cost, rtol=1e-13)
    oq, ot, oc, oit, ost = po.solve_batch(mode, offsets, f1, f2, None if c2 is None else po.covs_to_colmajor9(c2),
                                          None if c1 is None else po.covs_to_colmajor9(c1), 1e-13, q0, t0,
                                          options=po.default_options(jacobian_mode=po.JAC_NUMERIC_CENTRAL))

    def ang(a, bq):
        d = np.abs(np.sum(a * bq, axis=1)).clip(0, 1)
        return 2.0 * np.arccos(d)
    np.testing.assert_array_equal(rn.iterations, oit)
    np.testing.assert_array_equal(rn.status, ost)
    worst = max(math.radians(po.rotational_difference_deg(po.rot_from_quat(rn.q[p]), po.rot_from_quat(oq[p]))) for p in range(B))
    assert worst <= 1e-8, worst
    np.testing.assert_allclose(rn.cost, oc, rtol=1e-9)
    worst_a = max(math.radians(po.rotational_difference_deg(po.rot_from_quat(ra.q[p]), po.rot_from_quat(rn.q[p]))) for p in range(B))
    assert worst_a <= 1e-6, worst_a


@pytest.mark.gpu
def test_numeric_jacobian_mode_is_refused_where_it_is_not_built(oracle):
    from pnec_amd import capi
    from pnec_amd.streaming import Stream
    f1, f2, cov, q0, t0 = _zero_gradient_pair()
    with Stream(max_corr=64, slots=2) as st:
        with pytest.raises(Exception, match="numeric-Jacobian"):
            st.submit(capi.MODE_TARGET, f1, f2, cov, None, q0, t0,
                      options=capi.default_options(flags=capi.OPT_JACOBIAN_NUMERIC_CENTRAL))
    from pnec_amd import Batch
    with Batch(capi.MODE_TARGET, np.array([0, len(f1)], dtype=np.int64)) as b:
        b.fill(f1, f2, cov)
        with pytest.raises(Exception, match="undefined bit"):
            b.solve(q0[None], t0[None], options=capi.default_options(flags=4))
