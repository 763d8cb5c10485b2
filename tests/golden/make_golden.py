#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference's Python modules (this container only).

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box -> the .npz files are committed, this script is
only their provenance).  Nothing from the reference's source text is stored: the fixtures are
seeded inputs made here plus the numbers the reference's functions return for them.

Reference functions exercised (scripts/pnec/...):
  common.pnec_energy_rotations   common.py:13-37   <-> PNECResidualTarget (pnec_residual.h:86-104)
  common.nec_energy_rotations    common.py:40-59   <-> NECResidual (nec_residual.h:51-63)
  common.pnec_energy_translations common.py:62-86
  math.skew, math.unscented_transform (diagonal covariances only: for non-diagonal ones the
      Python uses ROWS of the Cholesky factor where the C++ uses COLUMNS -- SURVEY.md 8c)
  math.rotation_between_points   math.py:42-64     <-> RotationBetweenPoints (common.cc:118-124)
  math.unscented_transform(..., omnidirectional=True)  math.py:73-123: covariances that are DIAGONAL in the tangent
      frame of the point (cov = Rb diag(a, b, 0) Rb', Rb = rotation_between_points(z, point)) -- the same restriction
      as the pinhole case, for the same reason (rows vs columns of the 2x2 Cholesky factor)
  scf.fibonacci_sphere, scf.obj_fun   scf.py
"""
import os
import sys

import numpy as np

REF = "/root/reference/scripts"
HERE = os.path.dirname(os.path.abspath(__file__))


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def random_bearings(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def random_bearing_covs(rng, f, anisotropic):
    """rank-2-ish PSD covariances tangent to the bearing, magnitudes like pixel noise / f^2"""
    n = len(f)
    out = np.zeros((n, 3, 3))
    for i in range(n):
        a = np.cross(f[i], rng.normal(size=3))
        a /= np.linalg.norm(a)
        b = np.cross(f[i], a)
        if anisotropic:
            s1, s2 = rng.uniform(0.2, 3.0, size=2) * 1.5e-6
        else:
            s1 = s2 = 1.5e-6
        out[i] = s1 * np.outer(a, a) + s2 * np.outer(b, b) + 1e-12 * np.outer(f[i], f[i])
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be regenerated in the build container")
    sys.path.insert(0, REF)
    import pnec.common as rc   # noqa: E402  (the reference)
    import pnec.math as rm     # noqa: E402
    import pnec.scf as rs      # noqa: E402

    cases = {}
    idx = 0
    for seed in (1, 2, 3):
        for n in (10, 100, 512):
            for aniso in (False, True):
                rng = np.random.default_rng(1000 * seed + n + int(aniso))
                f1 = random_bearings(rng, n)
                f2 = random_bearings(rng, n)
                sig = random_bearing_covs(rng, f2, aniso)
                rots = np.stack([random_rotation(rng) for _ in range(4)]).reshape(2, 2, 3, 3)
                t = rng.normal(size=3)
                t /= np.linalg.norm(t)
                ts = rng.normal(size=(2, 2, 3))
                ts /= np.linalg.norm(ts, axis=-1, keepdims=True)
                for reg in (1e-13, 1e-10):
                    e_p = rc.pnec_energy_rotations(rots, t, f1, f2, sig, reg)
                    e_n = rc.nec_energy_rotations(rots, t, f1, f2)
                    e_t = rc.pnec_energy_translations(ts, rots[0, 0], f1, f2, sig, reg)
                    k = f"case{idx:03d}"
                    cases[k + "_f1"] = f1
                    cases[k + "_f2"] = f2
                    cases[k + "_sigmas"] = sig
                    cases[k + "_rotations"] = rots
                    cases[k + "_t"] = t
                    cases[k + "_ts"] = ts
                    cases[k + "_reg"] = np.array(reg)
                    cases[k + "_pnec_energy_rotations"] = e_p
                    cases[k + "_nec_energy_rotations"] = e_n
                    cases[k + "_pnec_energy_translations"] = e_t
                    idx += 1
    cases["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "energy_golden.npz"), **cases)

    # skew + unscented transform (pinhole, diagonal image covariances) + fibonacci sphere
    rng = np.random.default_rng(77)
    vs = rng.normal(size=(8, 3))
    skews = rm.skew(vs)
    pts = np.stack([rng.uniform(-400, 400, 16), rng.uniform(-300, 300, 16), np.full(16, 800.0)], 1)
    diag = rng.uniform(0.2, 2.0, size=(16, 2))
    covs = np.zeros((16, 3, 3))
    covs[:, 0, 0] = diag[:, 0]
    covs[:, 1, 1] = diag[:, 1]
    ut = np.stack([rm.unscented_transform(pts[i], covs[i], False, 1.0) for i in range(16)])
    # rotation_between_points + the omnidirectional branch of the unscented transform (round 4; a generator of their
    # own, so that the earlier fixtures keep their values)
    rng4 = np.random.default_rng(404)
    rb_p1 = random_bearings(rng4, 12)
    rb_p2 = random_bearings(rng4, 12)
    rb_p1[0], rb_p2[0] = np.array([0.0, 0.0, 1.0]), np.array([0.0, 0.0, 1.0])            # identity
    rb_p1[1], rb_p2[1] = np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0])            # a quarter turn
    rb_out = np.stack([rm.rotation_between_points(rb_p1[i], rb_p2[i]) for i in range(12)])
    omni_pts = random_bearings(rng4, 24) * rng4.uniform(4.0, 8.0, size=(24, 1))             # sphere of radius U(4, 8) (experiments.cc:109-126)
    omni_pts[:, 2] = np.abs(omni_pts[:, 2]) + 0.2                                        # away from the antipode of +z
    omni_pts[0] = np.array([0.0, 0.0, 5.0])                                              # on the axis: Rb = I
    omni_diag = rng4.uniform(0.2, 2.0, size=(24, 2)) * 1e-4
    omni_covs = np.zeros((24, 3, 3))
    for i in range(24):
        Rb = rm.rotation_between_points(np.array([0.0, 0.0, 1.0]), omni_pts[i] / np.linalg.norm(omni_pts[i]))
        omni_covs[i] = Rb @ np.diag([omni_diag[i, 0], omni_diag[i, 1], 0.0]) @ Rb.T
    omni_out = np.stack([rm.unscented_transform(omni_pts[i], omni_covs[i], True, 1.0) for i in range(24)])
    fib = np.asarray(rs.fibonacci_sphere(500))
    # scf.obj_fun (hard-wired to k = 10 matrices): sum_i x'A_i x / x'B_i x
    nvec = rng.normal(size=(10, 3))
    Ai = nvec[:, :, None] * nvec[:, None, :]
    Lb = rng.normal(size=(10, 3, 3)) * 1e-3
    Bi = Lb @ np.transpose(Lb, (0, 2, 1)) + 1e-9 * np.eye(3)
    X = rng.normal(size=(6, 3))
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    obj = rs.obj_fun(X, Ai, Bi)
    np.savez_compressed(os.path.join(HERE, "math_golden.npz"), skew_in=vs, skew_out=skews,
                        ut_points=pts, ut_covs=covs, ut_out=ut, fibonacci_500=fib,
                        rb_p1=rb_p1, rb_p2=rb_p2, rb_out=rb_out,
                        omni_points=omni_pts, omni_covs=omni_covs, omni_diag=omni_diag, omni_out=omni_out,
                        obj_Ai=Ai, obj_Bi=Bi, obj_X=X, obj_out=obj)
    print(f"wrote {idx} energy cases, math goldens")

    # ---- HOST and SYMMETRIC residuals (pnec_residual.h:50-83, :106-150), pinned through the reference's TARGET energy ----
    # The reference's Python has no Host / Symmetrical energy.  But its target energy evaluates
    #     (t' [fi]x R fi')^2 / (t' [fi]x R S R' [fi]x' t + reg)
    # for whatever it is given, and the C++ Host functor's denominator is exactly that form at fi := R f1, R := I, S := cov
    # (pnec_residual.h:63-70: bv_1_hat = Skew(R bv_1); bv_1_hat cov bv_1_hat'), the Symmetrical functor's second term the
    # same at fi := R f2, S := cov_1 (:128-140).  The numerator of all three functors is t . (f1 x R f2); the vector handed
    # over as fi' is chosen so that the Python's numerator t' [fi]x fi' equals it: fi' = n (t x fi) / |t x fi|^2.  So:
    #   host energy   = pnec_energy_rotations(I, t, R f1, fi', cov, reg)                      (one call over all k)
    #   symmetric r^2 = N / (N / E_target + N / E_second - reg), per correspondence, with N = nec_energy_rotations
    #                   (the squared numerator), E_target and E_second the Python's one-correspondence energies
    # -- every number on the right is what a function of scripts/pnec/common.py returned.
    forms = {}
    fidx = 0
    eye = np.eye(3).reshape(1, 1, 3, 3)
    for seed in (11, 12):
        for k in (12, 40):
            rng = np.random.default_rng(5000 + 10 * seed + k)
            f1 = random_bearings(rng, k)
            f2 = random_bearings(rng, k)
            cov2 = random_bearing_covs(rng, f2, True)      # frame-2 covariances (target term)
            cov1 = random_bearing_covs(rng, f1, True)      # frame-1 covariances (host functor / symmetric second term)
            for reg in (1e-13, 1e-10):
                for _pose in range(3):
                    R = random_rotation(rng)
                    t = rng.normal(size=3)
                    t /= np.linalg.norm(t)
                    n = np.einsum("i,ki->k", t, np.cross(f1, f2 @ R.T))          # t . (f1 x R f2)
                    # HOST: cov = the single covariance array of Optimize(..., frame = Host): frame 1's
                    a = f1 @ R.T
                    ta = np.cross(t, a)
                    x = n[:, None] * ta / np.sum(ta * ta, axis=1, keepdims=True)
                    host = float(rc.pnec_energy_rotations(eye, t, a, x, cov1, reg)[0, 0])
                    # SYMMETRIC, per correspondence
                    b = f2 @ R.T
                    tb = np.cross(t, b)
                    xb = n[:, None] * tb / np.sum(tb * tb, axis=1, keepdims=True)
                    r2 = np.zeros(k)
                    for i in range(k):
                        N = float(rc.nec_energy_rotations(R.reshape(1, 1, 3, 3), t, f1[i:i + 1], f2[i:i + 1])[0, 0])
                        Et = float(rc.pnec_energy_rotations(R.reshape(1, 1, 3, 3), t, f1[i:i + 1], f2[i:i + 1], cov2[i:i + 1], reg)[0, 0])
                        Es = float(rc.pnec_energy_rotations(eye, t, b[i:i + 1], xb[i:i + 1], cov1[i:i + 1], reg)[0, 0])
                        r2[i] = N / (N / Et + N / Es - reg)
                    key = f"form{fidx:03d}"
                    forms[key + "_f1"], forms[key + "_f2"] = f1, f2
                    forms[key + "_cov1"], forms[key + "_cov2"] = cov1, cov2
                    forms[key + "_R"], forms[key + "_t"], forms[key + "_reg"] = R, t, np.array(reg)
                    forms[key + "_host_energy"] = np.array(host)
                    forms[key + "_sym_r2"] = r2
                    fidx += 1
    forms["n_cases"] = np.array(fidx)
    np.savez_compressed(os.path.join(HERE, "residual_forms_golden.npz"), **forms)
    print(f"wrote {fidx} host / symmetric residual cases")


if __name__ == "__main__":
    main()
