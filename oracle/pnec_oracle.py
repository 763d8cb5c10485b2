"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/pnec_oracle.h).  Nothing under pnec_amd/ does.

Also holds ``energy_numpy``: an independent numpy restatement of the reference's energy
(scripts/pnec/common.py:13-59 <-> include/optimization/pnec_residual.h:97-102), used to
cross-check the C restatement against the golden vectors.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpnec_oracle.so")

MODE_NEC, MODE_TARGET, MODE_HOST, MODE_SYM = 0, 1, 2, 3
JAC_NUMERIC_CENTRAL, JAC_ANALYTIC = 0, 1
TERM_NAMES = {
    0: "function_tolerance",
    1: "parameter_tolerance",
    2: "gradient_tolerance",
    3: "max_iterations",
    4: "min_trust_region_radius",
    5: "invalid_steps",
    6: "bad_initial_point",
}


class Options(C.Structure):
    """Mirror of ``pnec_oracle_options`` (ceres::Solver::Options subset, Ceres 2.x defaults)."""

    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("check_convergence", C.c_int32),
        ("jacobian_mode", C.c_int32),
        ("reserved", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile oracle/libpnec_oracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                          for f in ("pnec_oracle.c", "pnec_oracle_frontend.c", "pnec_oracle_opengv.c", "pnec_oracle.h"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "libpnec_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def use_library(path: str | None = None) -> str:
    """Load another build of the same sources from here on (bench.py: the -march=native build made on the box the CPU
    baseline is timed on); None: back to oracle/libpnec_oracle.so.  Returns the path in use."""
    global _lib, _LIB_PATH
    _LIB_PATH = path or os.path.join(_HERE, "libpnec_oracle.so")
    _lib = None
    lib()
    return _LIB_PATH


def build_native() -> str | None:
    """`make -C oracle native` (gcc -O3 -march=native on THIS host); None when there is no compiler or the build fails."""
    import shutil
    if not (shutil.which("gcc") or shutil.which("cc")) or not shutil.which("make"):
        return None
    try:
        subprocess.run(["make", "-C", _HERE, "native"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=300)
    except (subprocess.SubprocessError, OSError):
        return None
    out = os.path.join(_HERE, "_native", "libpnec_oracle.so")
    return out if os.path.exists(out) else None


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.pnec_oracle_default_options.argtypes = [C.POINTER(Options)]
        _lib.pnec_oracle_angles_from_vec.argtypes = [_dp, _dp, _dp]
        _lib.pnec_oracle_quat_from_rot.argtypes = [_dp, _dp]
        _lib.pnec_oracle_rot_from_quat.argtypes = [_dp, _dp]
        _lib.pnec_oracle_result.argtypes = [_dp, C.c_double, C.c_double, _dp, _dp]
        _lib.pnec_oracle_rotational_difference_deg.argtypes = [_dp, _dp]
        _lib.pnec_oracle_rotational_difference_deg.restype = C.c_double
        _lib.pnec_oracle_translational_difference_deg.argtypes = [_dp, _dp, C.c_int]
        _lib.pnec_oracle_translational_difference_deg.restype = C.c_double
        _lib.pnec_oracle_cost_function.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, _dp]
        _lib.pnec_oracle_cost_function.restype = C.c_double
        _lib.pnec_oracle_residual.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_double,
                                              C.c_double, C.c_double, _dp]
        _lib.pnec_oracle_residual.restype = C.c_double
        _lib.pnec_oracle_energy.argtypes = [C.c_int, C.c_int64, _dp, _dp, _dp, _dp, C.c_double,
                                            _dp, _dp]
        _lib.pnec_oracle_energy.restype = C.c_double
        _lib.pnec_oracle_evaluate.argtypes = [C.c_int, C.c_int, C.c_int64, _dp, _dp, _dp, _dp,
                                              C.c_double, C.c_double, C.c_double, _dp, _dp, _dp,
                                              _dp]
        _lib.pnec_oracle_solve.argtypes = [C.c_int, C.c_int64, _dp, _dp, _dp, _dp, C.c_double, _dp,
                                           _dp, C.POINTER(Options), _dp, _dp, _dp, _dp, _ip]
        _lib.pnec_oracle_solve.restype = C.c_int
        _lib.pnec_oracle_solve_batch.argtypes = [C.c_int, C.c_int64, _lp, _dp, _dp, _dp, _dp,
                                                 C.c_double, _dp, _dp, C.c_int32, _dp,
                                                 C.POINTER(Options), C.c_int, _dp, _dp, _dp, _ip,
                                                 _ip]
        _lib.pnec_oracle_max_threads.restype = C.c_int
        _lib.pnec_oracle_lm_invalid_steps.argtypes = [C.c_int]
        _lib.pnec_oracle_lm_invalid_steps.restype = C.c_longlong
        _lib.pnec_oracle_lm_diagnostics.argtypes = [C.c_int]
        _lib.pnec_oracle_set_numeric_step_scale.argtypes = [C.c_double]
        _lib.pnec_oracle_rotation_between_points.argtypes = [_dp, _dp, _dp]
        _lib.pnec_oracle_rotation_between_points.restype = None
        _lib.pnec_oracle_unscented_transform.argtypes = [_dp, _dp, _dp, C.c_double, C.c_int, _dp]
        _lib.pnec_oracle_unproject.argtypes = [_dp, _dp, _dp]
        _lib.pnec_oracle_sym_eig3.argtypes = [_dp, _dp, _dp]
        _lib.pnec_oracle_compose_m.argtypes = [C.c_int64, _dp, _dp, _dp, C.c_int, _dp]
        _lib.pnec_oracle_translation_from_m.argtypes = [_dp, _dp]
        _lib.pnec_oracle_weight.argtypes = [_dp, _dp, _dp, _dp, _dp, C.c_double, C.c_int]
        _lib.pnec_oracle_weight.restype = C.c_double
        _lib.pnec_oracle_cayley_to_rot.argtypes = [_dp, _dp]
        _lib.pnec_oracle_rot_to_cayley.argtypes = [_dp, _dp]
        _lib.pnec_oracle_eigensolver.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, _ip]
        _lib.pnec_oracle_fibonacci_sphere.argtypes = [C.c_int, _dp]
        _lib.pnec_oracle_obj_fun.argtypes = [_dp, C.c_int64, _dp, _dp]
        _lib.pnec_oracle_obj_fun.restype = C.c_double
        _lib.pnec_oracle_scf.argtypes = [C.c_int64, _dp, _dp, _dp, C.c_int, _dp]
        _lib.pnec_oracle_build_ab.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp]
        _lib.pnec_oracle_nec_eigensolver.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, _dp]
        _lib.pnec_oracle_rng_uniform.argtypes = [C.c_uint64] * 4
        _lib.pnec_oracle_rng_uniform.restype = C.c_double
        _lib.pnec_oracle_reprojection_score.argtypes = [_dp, _dp, _dp, _dp]
        _lib.pnec_oracle_reprojection_score.restype = C.c_double
        _lib.pnec_oracle_ransac_eigensolver.argtypes = [C.c_int64, _dp, _dp, _dp, C.c_uint64, C.c_uint64, C.c_int,
                                                        C.c_int, C.c_double, _dp, _dp, C.POINTER(C.c_uint8), _ip, _ip]
        _lib.pnec_oracle_weighted_eigensolver.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                                          C.c_int, _dp, _dp]
        _lib.pnec_oracle_weighted_eigensolver_batch.argtypes = [C.c_int64, _lp, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                                                C.c_int, C.c_int, C.c_int, _dp, _dp]
        _lib.pnec_oracle_weighted_eigensolver_ex.argtypes = [C.c_int64, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                                             C.c_int, C.c_int, _dp, _dp]
        _u8 = C.POINTER(C.c_uint8)
        _lib.pnec_oracle_solve_chain_batch.argtypes = [C.c_int64, _lp, _dp, _dp, _dp, _dp, C.c_uint64, C.c_uint64,
                                                       C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                                       _dp, _dp, _u8, _ip, _ip, _dp, _dp, _dp, _dp, _ip, _ip]
    return _lib


def _d(a):
    """contiguous float64 view + pointer (None -> NULL)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def default_options(**overrides) -> Options:
    o = Options()
    lib().pnec_oracle_default_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def covs_to_colmajor9(covs: np.ndarray) -> np.ndarray:
    """[n,3,3] -> [n,9] in Eigen column-major order (how std::vector<Matrix3d> stores them)."""
    covs = np.asarray(covs, dtype=np.float64)
    return np.ascontiguousarray(np.transpose(covs, (0, 2, 1)).reshape(-1, 9))


CAMERA_OMNIDIRECTIONAL, CAMERA_PINHOLE = 0, 1


def rotation_between_points(p1, p2):
    """common.cc:118-124 (unit vectors in, 3x3 numpy matrix out: p2 = R p1)."""
    a, ap = _d(p1)
    b, bp = _d(p2)
    out = np.zeros(9)
    lib().pnec_oracle_rotation_between_points(ap, bp, out.ctypes.data_as(_dp))
    return out.reshape(3, 3).T


def unscented_transform(mu, cov, K_inv=None, kappa=1.0, camera_model=CAMERA_PINHOLE):
    """common.cc:467-525 for one point; cov / K_inv / result are 3x3 numpy matrices."""
    K_inv = np.eye(3) if K_inv is None else np.asarray(K_inv, dtype=np.float64)
    m, mp = _d(mu)
    c, cp = _d(np.asarray(cov, dtype=np.float64).T.reshape(9))
    k, kp = _d(K_inv.T.reshape(9))
    out = np.zeros(9)
    lib().pnec_oracle_unscented_transform(mp, cp, kp, float(kappa), int(camera_model),
                                          out.ctypes.data_as(_dp))
    return out.reshape(3, 3).T


def unproject(img_pt, K_inv):
    p, pp = _d(img_pt)
    k, kp = _d(np.asarray(K_inv, dtype=np.float64).T.reshape(9))
    out = np.zeros(3)
    lib().pnec_oracle_unproject(pp, kp, out.ctypes.data_as(_dp))
    return out


# ---- stages in front of the refinement (pnec_oracle_frontend.c) -------------------------------
def sym_eig3(A):
    a, ap = _d(np.asarray(A).reshape(9))
    w, V = np.zeros(3), np.zeros(9)
    lib().pnec_oracle_sym_eig3(ap, w.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return w, V.reshape(3, 3)


def compose_m(bvs1, bvs2, R, skip_first=True):
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    r, rp = _d(np.asarray(R).reshape(9))
    M = np.zeros(9)
    lib().pnec_oracle_compose_m(len(b1), b1p, b2p, rp, int(skip_first), M.ctypes.data_as(_dp))
    return M.reshape(3, 3)


def translation_from_m(M):
    m, mp = _d(np.asarray(M).reshape(9))
    t = np.zeros(3)
    lib().pnec_oracle_translation_from_m(mp, t.ctypes.data_as(_dp))
    return t


def weight(f1, f2, t, R, cov, reg, host_frame=False):
    a, ap = _d(f1); b, bp = _d(f2); c, cp = _d(t)
    r, rp = _d(np.asarray(R).reshape(9))
    s, sp = _d(np.asarray(cov).T.reshape(9))
    return lib().pnec_oracle_weight(ap, bp, cp, rp, sp, reg, int(host_frame))


def cayley_to_rot(v):
    a, ap = _d(v)
    R = np.zeros(9)
    lib().pnec_oracle_cayley_to_rot(ap, R.ctypes.data_as(_dp))
    return R.reshape(3, 3)


def rot_to_cayley(R):
    r, rp = _d(np.asarray(R).reshape(9))
    v = np.zeros(3)
    lib().pnec_oracle_rot_to_cayley(rp, v.ctypes.data_as(_dp))
    return v


def set_eigensolver_scheme(scheme: int) -> None:
    """0 (default): the damped Newton iteration the device runs by default; 1, 2: [EXT, from memory, unpinned] the two
    recollections of opengv's own iteration (pnec_oracle_opengv.c) -- 1 the normalised steepest descent with an adaptive
    step, which stops ~1e-5 rad short of the minimiser; 2 Eigen's Levenberg-Marquardt on the gradient of lambda_min
    composed with the reduced Cayley rotation.  Process-wide; affects every eigenvalue minimisation of the front stages."""
    L = lib()
    L.pnec_oracle_set_eigensolver_scheme.argtypes = [C.c_int]
    L.pnec_oracle_set_eigensolver_scheme.restype = None
    L.pnec_oracle_set_eigensolver_scheme(int(scheme))


def set_ransac_chained_starts(on: bool) -> None:
    """RANSAC hypothesis h + 1 starts from the last SCORED model's rotation + jitter (opengv's
    EigensolverSacProblem::getSelectedDistancesToModel leaves the scored model in the adapter [EXT, recalled]);
    off by default.  Process-wide switch, test tooling."""
    L = lib()
    L.pnec_oracle_set_ransac_chained_starts.argtypes = [C.c_int]
    L.pnec_oracle_set_ransac_chained_starts.restype = None
    L.pnec_oracle_set_ransac_chained_starts(int(bool(on)))


def set_ransac_frozen_rules(on: bool) -> None:
    """RANSAC under scheme 0 with the round-3 rules: every hypothesis scored, 50 iterations for a hypothesis' minimisation
    (pnec_oracle_frontend.c); OFF by default"""
    L = lib()
    L.pnec_oracle_set_ransac_frozen_rules.argtypes = [C.c_int]
    L.pnec_oracle_set_ransac_frozen_rules.restype = None
    L.pnec_oracle_set_ransac_frozen_rules(int(bool(on)))


def set_eigensolver_restart(on: bool) -> None:
    """scheme 1 only: ge_main2's disturbed-restart loop around the descent (off by default; pnec_oracle_opengv.c)"""
    L = lib()
    L.pnec_oracle_set_eigensolver_restart.argtypes = [C.c_int]
    L.pnec_oracle_set_eigensolver_restart(1 if on else 0)


def sums36(bvs1, bvs2):
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    G = np.zeros(36)
    lib().pnec_oracle_sums36.argtypes = [C.c_int64, _dp, _dp, _dp]
    lib().pnec_oracle_sums36(len(b1), b1p, b2p, G.ctypes.data_as(_dp))
    return G


def es_value_grad_sums(G, v, reduced=False):
    """-> (lambda_min, gradient [3], eigenvector [3], second eigenvalue) of M(v) composed from the 36 sums"""
    g_, gp = _d(G)
    v_, vp = _d(v)
    g, e, ev2 = np.zeros(3), np.zeros(3), C.c_double()
    L = lib()
    L.pnec_oracle_es_value_grad_sums.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.POINTER(C.c_double)]
    L.pnec_oracle_es_value_grad_sums.restype = C.c_double
    lam = L.pnec_oracle_es_value_grad_sums(gp, vp, 1 if reduced else 0, g.ctypes.data_as(_dp), e.ctypes.data_as(_dp),
                                           C.byref(ev2))
    return lam, g, e, ev2.value


def es_descent(G, v0):
    """scheme 1 on the 36 sums -> (v, iterations, trips)"""
    g_, gp = _d(G)
    v = np.array(v0, dtype=np.float64)
    trips = C.c_int()
    L = lib()
    L.pnec_oracle_es_descent.argtypes = [_dp, _dp, C.POINTER(C.c_int)]
    it = L.pnec_oracle_es_descent(gp, v.ctypes.data_as(_dp), C.byref(trips))
    return v, it, trips.value


def es_lm(G, v0):
    """scheme 2 on the 36 sums -> (v, successful iterations, nfev, info)"""
    g_, gp = _d(G)
    v = np.array(v0, dtype=np.float64)
    nfev, info = C.c_int(), C.c_int()
    L = lib()
    L.pnec_oracle_es_lm.argtypes = [_dp, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    it = L.pnec_oracle_es_lm(gp, v.ctypes.data_as(_dp), C.byref(nfev), C.byref(info))
    return v, it, nfev.value, info.value


def eigensolver(bvs1, bvs2, R0):
    """rotation minimising the smallest eigenvalue of M(R) (Kneip-Lynen), started at R0"""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    r, rp = _d(np.asarray(R0).reshape(9))
    R = np.zeros(9)
    it = C.c_int32()
    lib().pnec_oracle_eigensolver(len(b1), b1p, b2p, rp, R.ctypes.data_as(_dp), C.byref(it))
    return R.reshape(3, 3), it.value


def fibonacci_sphere(samples=500):
    pts = np.zeros((samples, 3))
    lib().pnec_oracle_fibonacci_sphere(samples, pts.ctypes.data_as(_dp))
    return pts


def obj_fun(t, Ai, Bi):
    a, ap = _d(t)
    A, Ap = _d(np.asarray(Ai).reshape(-1, 9))
    B, Bp = _d(np.asarray(Bi).reshape(-1, 9))
    return lib().pnec_oracle_obj_fun(ap, len(A), Ap, Bp)


def scf(Ai, Bi, t0, steps=10):
    A, Ap = _d(np.asarray(Ai).reshape(-1, 9))
    B, Bp = _d(np.asarray(Bi).reshape(-1, 9))
    a, ap = _d(t0)
    t = np.zeros(3)
    lib().pnec_oracle_scf(len(A), Ap, Bp, ap, steps, t.ctypes.data_as(_dp))
    return t


def build_ab(bvs1, bvs2, covs, R, reg):
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c, cp = _d(covs_to_colmajor9(covs))
    r, rp = _d(np.asarray(R).reshape(9))
    n = len(b1)
    Ai, Bi = np.zeros((n, 3, 3)), np.zeros((n, 3, 3))
    lib().pnec_oracle_build_ab(n, b1p, b2p, cp, rp, reg, Ai.ctypes.data_as(_dp), Bi.ctypes.data_as(_dp))
    return Ai, Bi


def nec_eigensolver(bvs1, bvs2, R0):
    """PNEC::Eigensolver without RANSAC (pnec.cc:273-278) -> (R, t)"""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    r, rp = _d(np.asarray(R0).reshape(9))
    R, t = np.zeros(9), np.zeros(3)
    lib().pnec_oracle_nec_eigensolver(len(b1), b1p, b2p, rp, R.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
    return R.reshape(3, 3), t


def ransac_eigensolver(bvs1, bvs2, R0, seed=1, pair_id=0, max_iterations=5000, sample_size=10, threshold=1e-6):
    """PNEC::Eigensolver with use_ransac_ (pnec.cc:239-272) -> (R, t, inlier_mask, ransac_iterations)"""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    r, rp = _d(np.asarray(R0).reshape(9))
    n = len(b1)
    R, t = np.zeros(9), np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    cnt, it = C.c_int32(), C.c_int32()
    lib().pnec_oracle_ransac_eigensolver(n, b1p, b2p, rp, seed, pair_id, max_iterations, sample_size, threshold,
                                         R.ctypes.data_as(_dp), t.ctypes.data_as(_dp),
                                         mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(cnt), C.byref(it))
    return R.reshape(3, 3), t, mask[:n].astype(bool), it.value


def reprojection_score(f1, f2, R, t):
    a, ap = _d(f1); b, bp = _d(f2); r, rp = _d(np.asarray(R).reshape(9)); c, cp = _d(t)
    return lib().pnec_oracle_reprojection_score(ap, bp, rp, cp)


def weighted_eigensolver(bvs1, bvs2, covs, R_init, t_init, reg=1e-13, weighted_iterations=10,
                         device_early_exits=False):
    """PNEC::WeightedEigensolver (pnec.cc:283-348) -> (R, t).  Literal by default (every round re-runs
    the eigensolver, every scf call runs its 10 steps).  device_early_exits=True is NOT the reference:
    it takes the device kernel's two declared early exits (see pnec_oracle_frontend.c)."""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c, cp = _d(covs_to_colmajor9(covs))
    r, rp = _d(np.asarray(R_init).reshape(9))
    t0, t0p = _d(t_init)
    R, t = np.zeros(9), np.zeros(3)
    lib().pnec_oracle_weighted_eigensolver_ex(len(b1), b1p, b2p, cp, rp, t0p, reg, weighted_iterations,
                                              1 if device_early_exits else 0,
                                              R.ctypes.data_as(_dp), t.ctypes.data_as(_dp))
    return R.reshape(3, 3), t


def weighted_eigensolver_batch(offsets, bvs1, bvs2, covs, R_init, t_init, reg=1e-13, weighted_iterations=10,
                               device_early_exits=False, num_threads=0):
    """weighted_eigensolver for a ragged batch (OpenMP over pairs) -> (R [P,3,3], t [P,3])"""
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    P = len(off) - 1
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c, cp = _d(covs_to_colmajor9(covs))
    r, rp = _d(np.asarray(R_init).reshape(P, 9))
    t0, t0p = _d(np.asarray(t_init).reshape(P, 3))
    R, t = np.zeros((P, 9)), np.zeros((P, 3))
    lib().pnec_oracle_weighted_eigensolver_batch(P, off.ctypes.data_as(_lp), b1p, b2p, cp, rp, t0p, reg,
                                                 weighted_iterations, 1 if device_early_exits else 0,
                                                 num_threads or max_threads(), R.ctypes.data_as(_dp),
                                                 t.ctypes.data_as(_dp))
    return R.reshape(P, 3, 3), t


def solve_chain_batch(offsets, bvs1, bvs2, covs, init_q, seed=1, first_pair_id=0, max_ransac_iterations=5000,
                      sample_size=10, threshold=1e-6, reg=1e-13, weighted_iterations=10, num_threads=0):
    """PNEC::Solve with the reference's default Options (pnec.cc:77-124) for a ragged batch, OpenMP over pairs:
    RANSAC eigensolver (pair p draws as pair_id first_pair_id + p) -> InlierExtraction -> WeightedEigensolver
    (literal) -> CeresSolver (central differences, default solver options).  covs [M,3,3].
    -> dict(es_q, es_t, mask [M] bool, inlier_count, ransac_iterations, w_q, w_t, q, t, ls_iterations, ls_status)"""
    off = np.ascontiguousarray(offsets, dtype=np.int64)
    P, M = len(off) - 1, int(off[-1])
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c, cp = _d(covs_to_colmajor9(covs))
    q0, q0p = _d(np.asarray(init_q).reshape(P, 4))
    f = lambda *shape: np.zeros(shape)
    i = lambda n: np.zeros(n, dtype=np.int32)
    es_q, es_t, w_q, w_t, q, t = f(P, 4), f(P, 3), f(P, 4), f(P, 3), f(P, 4), f(P, 3)
    mask = np.zeros(max(M, 1), dtype=np.uint8)
    cnt, rit, lit, lst = i(P), i(P), i(P), i(P)
    dp = lambda a: a.ctypes.data_as(_dp)
    ip = lambda a: a.ctypes.data_as(_ip)
    lib().pnec_oracle_solve_chain_batch(P, off.ctypes.data_as(_lp), b1p, b2p, cp, q0p, seed, first_pair_id,
                                        max_ransac_iterations, sample_size, threshold, reg, weighted_iterations,
                                        num_threads or usable_threads(), dp(es_q), dp(es_t),
                                        mask.ctypes.data_as(C.POINTER(C.c_uint8)), ip(cnt), ip(rit), dp(w_q), dp(w_t),
                                        dp(q), dp(t), ip(lit), ip(lst))
    return dict(es_q=es_q, es_t=es_t, mask=mask[:M].astype(bool), inlier_count=cnt, ransac_iterations=rit, w_q=w_q,
                w_t=w_t, q=q, t=t, ls_iterations=lit, ls_status=lst)


def angles_from_vec(v):
    v_, vp = _d(v)
    th, ph = C.c_double(), C.c_double()
    lib().pnec_oracle_angles_from_vec(vp, C.byref(th), C.byref(ph))
    return th.value, ph.value


def quat_from_rot(R):
    R_, Rp = _d(np.asarray(R).reshape(9))
    q = np.zeros(4)
    lib().pnec_oracle_quat_from_rot(Rp, q.ctypes.data_as(_dp))
    return q


def rot_from_quat(q):
    q_, qp = _d(q)
    R = np.zeros(9)
    lib().pnec_oracle_rot_from_quat(qp, R.ctypes.data_as(_dp))
    return R.reshape(3, 3)


def rotational_difference_deg(R1, R2) -> float:
    a, ap = _d(np.asarray(R1).reshape(9))
    b, bp = _d(np.asarray(R2).reshape(9))
    return lib().pnec_oracle_rotational_difference_deg(ap, bp)


def translational_difference_deg(t1, t2, both_directions=True) -> float:
    a, ap = _d(t1)
    b, bp = _d(t2)
    return lib().pnec_oracle_translational_difference_deg(ap, bp, int(both_directions))


def cost_function(bvs1, bvs2, covs, R, t) -> float:
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c, cp = _d(covs_to_colmajor9(covs))
    R_, Rp = _d(np.asarray(R).reshape(9))
    t_, tp = _d(t)
    return lib().pnec_oracle_cost_function(len(b1), b1p, b2p, cp, Rp, tp)


def residual(mode, f1, f2, cov2, cov1, reg, theta, phi, q) -> float:
    f1_, f1p = _d(f1)
    f2_, f2p = _d(f2)
    c2, c2p = _d(None if cov2 is None else np.asarray(cov2).T.reshape(9))
    c1, c1p = _d(None if cov1 is None else np.asarray(cov1).T.reshape(9))
    q_, qp = _d(q)
    return lib().pnec_oracle_residual(mode, f1p, f2p, c2p, c1p, reg, theta, phi, qp)


def energy(mode, bvs1, bvs2, covs2, covs1, reg, R, t) -> float:
    """sum_i r_i^2 at an explicit (R, t) -- the quantity scripts/pnec/common.py computes."""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c2, c2p = _d(None if covs2 is None else covs_to_colmajor9(covs2))
    c1, c1p = _d(None if covs1 is None else covs_to_colmajor9(covs1))
    R_, Rp = _d(np.asarray(R).reshape(9))
    t_, tp = _d(t)
    return lib().pnec_oracle_energy(mode, len(b1), b1p, b2p, c2p, c1p, reg, Rp, tp)


def evaluate(mode, jacobian_mode, bvs1, bvs2, covs2, covs1, reg, theta, phi, q):
    """-> (r[n], J[n,5], cost=1/2 sum r^2) in the Ceres tangent space (theta, phi, delta_xyz)."""
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c2, c2p = _d(None if covs2 is None else covs_to_colmajor9(covs2))
    c1, c1p = _d(None if covs1 is None else covs_to_colmajor9(covs1))
    q_, qp = _d(q)
    n = len(b1)
    r = np.zeros(n)
    J = np.zeros((n, 5))
    cost = C.c_double()
    lib().pnec_oracle_evaluate(mode, jacobian_mode, n, b1p, b2p, c2p, c1p, reg, theta, phi, qp,
                               r.ctypes.data_as(_dp), J.ctypes.data_as(_dp), C.byref(cost))
    return r, J, cost.value


@dataclass
class Solution:
    q: np.ndarray          # xyzw, normalised
    t: np.ndarray          # unit
    R: np.ndarray          # 3x3
    theta: float
    phi: float
    cost: float            # 1/2 sum r^2 at the returned point
    iterations: int
    status: int


def solve(mode, bvs1, bvs2, covs2, covs1, reg, init_q, init_t, options: Options | None = None
          ) -> Solution:
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c2, c2p = _d(None if covs2 is None else covs_to_colmajor9(covs2))
    c1, c1p = _d(None if covs1 is None else covs_to_colmajor9(covs1))
    q0, q0p = _d(init_q)
    t0, t0p = _d(init_t)
    q = np.zeros(4)
    t = np.zeros(3)
    tp = np.zeros(2)
    cost = C.c_double()
    it = C.c_int32()
    st = lib().pnec_oracle_solve(mode, len(b1), b1p, b2p, c2p, c1p, reg, q0p, t0p,
                                 C.byref(options) if options is not None else None,
                                 q.ctypes.data_as(_dp), t.ctypes.data_as(_dp),
                                 tp.ctypes.data_as(_dp), C.byref(cost), C.byref(it))
    return Solution(q=q, t=t, R=rot_from_quat(q), theta=tp[0], phi=tp[1], cost=cost.value,
                    iterations=it.value, status=st)


def solve_batch(mode, offsets, bvs1, bvs2, covs2_9, covs1_9, reg, init_q, init_t,
                n_hyp=1, hyp_t=None, options: Options | None = None, num_threads=0):
    """Batch driver. covs*_9 are [sumN, 9] Eigen column-major (use covs_to_colmajor9)."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    B = len(offsets) - 1
    b1, b1p = _d(bvs1)
    b2, b2p = _d(bvs2)
    c2, c2p = _d(covs2_9)
    c1, c1p = _d(covs1_9)
    q0, q0p = _d(init_q)
    t0, t0p = _d(init_t)
    h, hp = _d(hyp_t)
    S = B * max(1, n_hyp)
    out_q = np.zeros((S, 4))
    out_t = np.zeros((S, 3))
    out_cost = np.zeros(S)
    out_it = np.zeros(S, dtype=np.int32)
    out_st = np.zeros(S, dtype=np.int32)
    lib().pnec_oracle_solve_batch(mode, B, offsets.ctypes.data_as(_lp), b1p, b2p, c2p, c1p, reg,
                                  q0p, t0p, n_hyp, hp,
                                  C.byref(options) if options is not None else None,
                                  num_threads, out_q.ctypes.data_as(_dp),
                                  out_t.ctypes.data_as(_dp), out_cost.ctypes.data_as(_dp),
                                  out_it.ctypes.data_as(_ip), out_st.ctypes.data_as(_ip))
    return out_q, out_t, out_cost, out_it, out_st


def max_threads() -> int:
    return lib().pnec_oracle_max_threads()


def usable_threads() -> int:
    """Threads worth starting: the OpenMP maximum capped by the affinity mask and by TWICE the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs, an OpenMP maximum of 128 and a quota of 16 CPUs: bench.py's scaling table there reads
    8 / 16 / 32 / 64 / 128 threads -> best at 32, 128 threads at less than half of that)."""
    n = max_threads()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, 2 * int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def lm_invalid_steps(reset: bool = True) -> int:
    """Invalid LM steps (Ceres' HandleInvalidStep branch) counted since the last reset; lm_diagnostics(True) first."""
    return int(lib().pnec_oracle_lm_invalid_steps(1 if reset else 0))


def lm_diagnostics(on: bool) -> None:
    lib().pnec_oracle_lm_diagnostics(1 if on else 0)


def set_numeric_step_scale(s: float) -> None:
    """Scale the central-difference step of the numeric Jacobian (1.0 = the reference's); see pnec_oracle.h."""
    lib().pnec_oracle_set_numeric_step_scale(float(s))


# --------------------------------------------------------------------------------------------
# Independent numpy restatement of the energies (no C involved).
def skew_numpy(v: np.ndarray) -> np.ndarray:
    """[...,3] -> [...,3,3]  (src/common/common.cc:96-101)"""
    v = np.asarray(v, dtype=np.float64)
    z = np.zeros_like(v[..., 0])
    return np.stack([
        np.stack([z, -v[..., 2], v[..., 1]], -1),
        np.stack([v[..., 2], z, -v[..., 0]], -1),
        np.stack([-v[..., 1], v[..., 0], z], -1),
    ], -2)


def energy_numpy(mode, bvs1, bvs2, covs2, covs1, reg, R, t) -> float:
    """sum_i r_i^2 with r_i per pnec_residual.h / nec_residual.h, evaluated in numpy."""
    f1 = np.asarray(bvs1, dtype=np.float64)
    f2 = np.asarray(bvs2, dtype=np.float64)
    R = np.asarray(R, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    Rf2 = f2 @ R.T
    num = np.cross(f1, Rf2) @ t
    if mode == MODE_NEC:
        return float(np.sum(num ** 2))
    den = np.full(len(f1), float(reg))
    F1 = skew_numpy(f1)
    if mode in (MODE_TARGET, MODE_SYM):
        v = np.einsum("i,nij,jk->nk", t, F1, R)
        den = den + np.einsum("ni,nij,nj->n", v, np.asarray(covs2), v)
    if mode == MODE_HOST:
        v = np.einsum("i,nij->nj", t, skew_numpy(f1 @ R.T))
        den = den + np.einsum("ni,nij,nj->n", v, np.asarray(covs2), v)
    if mode == MODE_SYM:
        v = np.einsum("i,nij->nj", t, skew_numpy(Rf2))
        den = den + np.einsum("ni,nij,nj->n", v, np.asarray(covs1), v)
    return float(np.sum(num ** 2 / den))
