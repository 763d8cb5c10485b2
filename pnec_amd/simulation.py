"""Synthetic frame pairs following the reference simulator's distributions (harness, not product).

What is reproduced (reference file:line):
  * relative pose: Euler roll/pitch/yaw ~ U(-0.5, 0.5) rad composed X*Y*Z; translation
    U(0,2) * uniform direction                      src/simulation/experiments.cc:43-70
  * pinhole points: depth ((1-0.4)U+0.4)*5, x in +-0.5, y in +-0.75 at unit depth, projected
    with f = 800; omnidirectional: uniform sphere, radius U(4,8)
                                                    src/simulation/experiments.cc:72-129
  * image-plane covariance noise*scale*Rot(a) diag(b,1-b) Rot(a)' with the four noise types
                                                    experiments.cc:174-185, standard_experiments.cc:85-126
  * noise added in frame 2 only, via the Cholesky factor     standard_experiments.cc:128-157
  * bearing covariances by the unscented transform, kappa = 1, K^-1 = I
                                                    src/common/common.cc:467-525, sim_common.cc:72-107
  * start pose = ground truth perturbed by <= 0.01 rad / <= 0.01, translation renormalised
                                                    src/simulation/sim_common.cc:205-231
The reference draws from libstdc++ mt19937 / default_random_engine; this generator uses torch's
Philox/MT streams with recorded seeds instead (bit-equality with the reference's draws is not a
goal; the distributions are).
Everything is vectorised over [B, N] and runs on CPU or GPU in float64.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

NOISE_TYPES = ("isotropic_homogeneous", "anisotropic_homogeneous", "isotropic_inhomogeneous",
               "anisotropic_inhomogeneous")


@dataclass
class SimBatch:
    bvs1: torch.Tensor    # [B,N,3] unit bearings, frame 1 (host)
    bvs2: torch.Tensor    # [B,N,3] unit bearings, frame 2 (target), noisy
    covs2: torch.Tensor   # [B,N,3,3] bearing covariances of frame 2
    R_gt: torch.Tensor    # [B,3,3]
    t_gt: torch.Tensor    # [B,3]   (not normalised: magnitude U(0,2))
    init_R: torch.Tensor  # [B,3,3]
    init_t: torch.Tensor  # [B,3]   unit
    init_q: torch.Tensor  # [B,4]   xyzw of init_R


def _rot_x(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([o, z, z, z, c, -s, z, s, c], -1).reshape(*a.shape, 3, 3)


def _rot_y(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([c, z, s, z, o, z, -s, z, c], -1).reshape(*a.shape, 3, 3)


def _rot_z(a):
    c, s, o, z = torch.cos(a), torch.sin(a), torch.ones_like(a), torch.zeros_like(a)
    return torch.stack([c, -s, z, s, c, z, z, z, o], -1).reshape(*a.shape, 3, 3)


def _sphere(u1, u2):
    """direction from two uniforms, as the reference: theta = 2 pi u1, phi = acos(1 - 2 u2)"""
    theta = 2 * math.pi * u1
    phi = torch.acos(1 - 2 * u2)
    return torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta),
                        torch.cos(phi)], -1)


def axis_angle_to_matrix(axis, angle):
    """Rodrigues; axis [...,3] unit, angle [...]"""
    x, y, z = axis[..., 0], axis[..., 1], axis[..., 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], -1).reshape(*x.shape, 3, 3)
    eye = torch.eye(3, dtype=axis.dtype, device=axis.device).expand(*x.shape, 3, 3)
    s = torch.sin(angle)[..., None, None]
    c = torch.cos(angle)[..., None, None]
    return eye + s * K + (1 - c) * (K @ K)


def matrix_to_quaternion_xyzw(R):
    """Eigen::Quaterniond(Matrix3d) (Shepperd), batched; returns xyzw."""
    m = R
    tr = m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]
    q = torch.zeros(*R.shape[:-2], 4, dtype=R.dtype, device=R.device)
    # branch tr > 0
    t = torch.sqrt(torch.clamp(tr, min=-0.999999) + 1.0)
    w0 = 0.5 * t
    f = 0.5 / t
    cand0 = torch.stack([(m[..., 2, 1] - m[..., 1, 2]) * f, (m[..., 0, 2] - m[..., 2, 0]) * f,
                         (m[..., 1, 0] - m[..., 0, 1]) * f, w0], -1)
    out = cand0.clone()
    diag = torch.stack([m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]], -1)
    i = torch.zeros_like(tr, dtype=torch.long)
    i = torch.where(diag[..., 1] > diag[..., 0], torch.ones_like(i), i)
    dii = torch.gather(diag, -1, i[..., None])[..., 0]
    i = torch.where(diag[..., 2] > dii, torch.full_like(i, 2), i)
    for ii in range(3):
        j, k = (ii + 1) % 3, (ii + 2) % 3
        tt = torch.sqrt(torch.clamp(m[..., ii, ii] - m[..., j, j] - m[..., k, k] + 1.0, min=1e-300))
        ff = 0.5 / tt
        c = torch.zeros_like(out)
        c[..., ii] = 0.5 * tt
        c[..., 3] = (m[..., k, j] - m[..., j, k]) * ff
        c[..., j] = (m[..., j, ii] + m[..., ii, j]) * ff
        c[..., k] = (m[..., k, ii] + m[..., ii, k]) * ff
        sel = ((tr <= 0) & (i == ii))[..., None]
        out = torch.where(sel, c, out)
    return out


def unscented_bearing_cov(mu, cov2d, kappa: float = 1.0):
    """Pinhole branch of pnec::common::UnscentedTransform with K^-1 = I (common.cc:467-525).

    mu    [...,3] image point (x, y, f)
    cov2d [...,2,2] image-plane covariance (top-left block of the reference's 3x3)
    ->    [...,3,3] covariance of the normalised bearing
    """
    n = 2
    a, b, d = cov2d[..., 0, 0], cov2d[..., 1, 0], cov2d[..., 1, 1]
    l00 = torch.sqrt(a)
    l10 = b / l00
    l11 = torch.sqrt(d - l10 * l10)
    zero = torch.zeros_like(a)
    col0 = torch.stack([l00, l10, zero], -1)  # columns of the lower Cholesky factor
    col1 = torch.stack([zero, l11, zero], -1)
    pts = torch.stack([mu, mu + col0, mu + col1, mu - col0, mu - col1], -2)  # [...,5,3]
    w0 = kappa / (n + kappa)
    wi = 0.5 / (n + kappa)
    w = torch.tensor([w0, wi, wi, wi, wi], dtype=mu.dtype, device=mu.device)
    tp = pts / torch.linalg.norm(pts, dim=-1, keepdim=True)
    mean = (w[:, None] * tp).sum(-2, keepdim=True)
    dlt = tp - mean
    return torch.einsum("k,...ki,...kj->...ij", w, dlt, dlt)


def unscented_bearing_cov_omni(mu, cov3, kappa: float = 1.0):
    """Omnidirectional branch of UnscentedTransform (common.cc:478-486,507-509).  cov3 [...,3,3]
    is the tangent-plane covariance already rotated to the bearing (as AddNoise stores it)."""
    n = 2
    v = mu / torch.linalg.norm(mu, dim=-1, keepdim=True)
    Rb = rotation_between_z_and(v)
    local = (Rb.transpose(-1, -2) @ cov3 @ Rb)[..., :2, :2]
    a, b, d = local[..., 0, 0], local[..., 1, 0], local[..., 1, 1]
    l00 = torch.sqrt(a)
    l10 = b / l00
    l11 = torch.sqrt(d - l10 * l10)
    zero = torch.zeros_like(a)
    c0 = (Rb @ torch.stack([l00, l10, zero], -1)[..., None])[..., 0]
    c1 = (Rb @ torch.stack([zero, l11, zero], -1)[..., None])[..., 0]
    pts = torch.stack([mu, mu + c0, mu + c1, mu - c0, mu - c1], -2)
    w0 = kappa / (n + kappa)
    wi = 0.5 / (n + kappa)
    w = torch.tensor([w0, wi, wi, wi, wi], dtype=mu.dtype, device=mu.device)
    tp = pts / torch.linalg.norm(pts, dim=-1, keepdim=True)
    mean = (w[:, None] * tp).sum(-2, keepdim=True)
    dlt = tp - mean
    return torch.einsum("k,...ki,...kj->...ij", w, dlt, dlt)


def rotation_between_z_and(v):
    """pnec::common::RotationBetweenPoints((0,0,1), v) (common.cc:118-124); v unit [...,3]"""
    z = torch.zeros_like(v)
    z[..., 2] = 1.0
    c = torch.cross(z, v, dim=-1)
    x, y, zz = c[..., 0], c[..., 1], c[..., 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -zz, y, zz, zero, -x, -y, x, zero], -1).reshape(*x.shape, 3, 3)
    dot = v[..., 2]
    eye = torch.eye(3, dtype=v.dtype, device=v.device).expand(*x.shape, 3, 3)
    return eye + K + (K @ K) / (1 + dot)[..., None, None]


def generate(n_pairs: int, n_corr: int, noise_type: str = "anisotropic_inhomogeneous",
             noise_level: float = 1.0, camera: str = "pinhole", seed: int = 1,
             device: str | torch.device = "cpu", init_scaling: float = 1.0,
             translation: bool = True) -> SimBatch:
    if noise_type not in NOISE_TYPES:
        raise ValueError(noise_type)
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    B, N = n_pairs, n_corr
    f64 = dict(dtype=torch.float64, device=dev)
    U = lambda *shape: torch.rand(*shape, generator=g, **f64)

    # --- poses (experiments.cc:43-70)
    e = (U(B, 3) * 2 - 1) * 0.5
    R_gt = _rot_x(e[:, 0]) @ _rot_y(e[:, 1]) @ _rot_z(e[:, 2])
    if translation:
        t_gt = (2 * U(B))[:, None] * _sphere(U(B), U(B))
    else:
        t_gt = torch.zeros(B, 3, **f64)

    # --- points (experiments.cc:72-129); pose_1 = identity, pose_2 = (R_gt, t_gt)
    focal = 800.0
    if camera == "pinhole":
        depth = ((1 - 0.4) * U(B, N) + 0.4) * 5.0
        P = torch.stack([(U(B, N) - 0.5) * 1.0, (U(B, N) - 0.5) * 1.5,
                         torch.ones(B, N, **f64)], -1) * depth[..., None]
        p1 = P / P[..., 2:3] * focal
        P2 = torch.einsum("bji,bnj->bni", R_gt, P - t_gt[:, None, :])  # R' (P - t)
        p2 = P2 / P2[..., 2:3] * focal
    elif camera == "omnidirectional":
        P = (4.0 * U(B, N) + 4.0)[..., None] * _sphere(U(B, N), U(B, N))
        p1 = P / torch.linalg.norm(P, dim=-1, keepdim=True) * focal
        P2 = torch.einsum("bji,bnj->bni", R_gt, P - t_gt[:, None, :])
        p2 = P2 / torch.linalg.norm(P2, dim=-1, keepdim=True) * focal
    else:
        raise ValueError(camera)

    # --- covariances (standard_experiments.cc:85-126, experiments.cc:174-185)
    if noise_type == "isotropic_homogeneous":
        alpha = torch.zeros(B, N, **f64); beta = torch.full((B, N), 0.5, **f64)
        scale = torch.ones(B, N, **f64)
    elif noise_type == "anisotropic_homogeneous":
        alpha = U(B, N) * math.pi
        beta = ((U(B, 1) + 1.0) / 2.0).expand(B, N)  # drawn once per experiment
        scale = torch.ones(B, N, **f64)
    elif noise_type == "isotropic_inhomogeneous":
        alpha = torch.zeros(B, N, **f64); beta = torch.full((B, N), 0.5, **f64)
        scale = U(B, N) + 0.5
    else:
        alpha = U(B, N) * math.pi
        beta = (U(B, N) + 1.0) / 2.0
        scale = U(B, N) + 0.5
    ca, sa = torch.cos(alpha), torch.sin(alpha)
    rot = torch.stack([ca, -sa, sa, ca], -1).reshape(B, N, 2, 2)
    dg = torch.zeros(B, N, 2, 2, **f64)
    dg[..., 0, 0] = beta
    dg[..., 1, 1] = 1.0 - beta
    cov2d = (noise_level * scale)[..., None, None] * (rot @ dg @ rot.transpose(-1, -2))

    # --- noise in frame 2 (standard_experiments.cc:128-157)
    a, b, d = cov2d[..., 0, 0], cov2d[..., 1, 0], cov2d[..., 1, 1]
    l00 = torch.sqrt(a); l10 = b / l00; l11 = torch.sqrt(d - l10 * l10)
    z = torch.randn(B, N, 2, generator=g, **f64)
    nx = l00 * z[..., 0]
    ny = l10 * z[..., 0] + l11 * z[..., 1]
    noise_local = torch.stack([nx, ny, torch.zeros_like(nx)], -1)
    if camera == "pinhole":
        p2n = p2 + noise_local
        covs2 = unscented_bearing_cov(p2n, cov2d)
    else:
        Rb = rotation_between_z_and(p2 / torch.linalg.norm(p2, dim=-1, keepdim=True))
        p2n = p2 + (Rb @ noise_local[..., None])[..., 0]
        cov3 = torch.zeros(B, N, 3, 3, **f64)
        cov3[..., :2, :2] = cov2d
        covs2 = unscented_bearing_cov_omni(p2n, Rb @ cov3 @ Rb.transpose(-1, -2))

    bvs1 = p1 / torch.linalg.norm(p1, dim=-1, keepdim=True)
    bvs2 = p2n / torch.linalg.norm(p2n, dim=-1, keepdim=True)

    # --- start pose near the ground truth (sim_common.cc:205-231)
    axis = _sphere(U(B), U(B))
    angle = torch.sqrt(U(B)) * 0.01 * init_scaling
    R_off = axis_angle_to_matrix(axis, angle)
    t_off = (torch.sqrt(U(B)) * 0.01 * init_scaling)[:, None] * _sphere(U(B), U(B))
    init_R = R_off @ R_gt
    init_t = torch.einsum("bij,bj->bi", R_off, t_gt) + t_off
    init_t = init_t / torch.linalg.norm(init_t, dim=-1, keepdim=True)
    init_q = matrix_to_quaternion_xyzw(init_R)
    return SimBatch(bvs1, bvs2, covs2, R_gt, t_gt, init_R, init_t, init_q)


def generate_kitti_like(n_pairs: int, mean_corr: int = 500, seed: int = 1,
                        device: str | torch.device = "cpu", counts=None, cov_model: str = "klt"):
    """KITTI-like SYNTHETIC stream (no KITTI data exists in this environment): forward motion
    (t ~ +z, the (theta,phi) chart's singular direction, Appendix C12), small yaw, pinhole
    fx = 718.856 on a 1241x376 image (data/config_kitti00-02.yaml:8-11), ragged track counts.
    Returns (offsets int64 [B+1] numpy, bvs1 [M,3], bvs2 [M,3], covs2 [M,3,3], R_gt, t_gt,
    init_q [B,4], init_t [B,3]).  counts: optional int64 [B] pair sizes (else drawn here).
    cov_model: "klt" (default since round 5) draws the 2x2 image covariances from the reference's KLT patch model,
    "simulator" from the simulator's anisotropic model (rounds 1-4)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    f64 = dict(dtype=torch.float64, device=dev)
    U = lambda *shape: torch.rand(*shape, generator=g, **f64)
    B = n_pairs
    if counts is None:
        counts = torch.clamp((mean_corr + 60.0 * torch.randn(B, generator=g, **f64)).round(),
                             min=64, max=mean_corr + 200).to(torch.int64)
    else:
        counts = torch.as_tensor(counts, dtype=torch.int64, device=dev)
        if counts.shape != (B,):
            raise ValueError("counts must be [n_pairs]")
    offsets = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    offsets[1:] = torch.cumsum(counts, 0)
    M = int(offsets[-1].item())
    pair = torch.repeat_interleave(torch.arange(B, device=dev), counts)
    fx, cx, cy, W, H = 718.856, 607.1928, 185.2157, 1241.0, 376.0
    yaw = (U(B) - 0.5) * 0.04
    pitch = (U(B) - 0.5) * 0.004
    R_gt = _rot_y(yaw) @ _rot_x(pitch)
    tdir = torch.stack([(U(B) - 0.5) * 0.04, (U(B) - 0.5) * 0.02, torch.ones(B, **f64)], -1)
    t_gt = tdir / torch.linalg.norm(tdir, dim=-1, keepdim=True) * (0.6 + 0.6 * U(B))[:, None]
    u = U(M) * W
    v = U(M) * H
    depth = 4.0 + 40.0 * U(M) ** 2
    P = torch.stack([(u - cx) / fx, (v - cy) / fx, torch.ones(M, **f64)], -1) * depth[:, None]
    P2 = torch.einsum("mji,mj->mi", R_gt[pair], P - t_gt[pair])
    p1 = torch.stack([(u - cx), (v - cy), torch.full((M,), fx, **f64)], -1)
    p2 = P2 / P2[:, 2:3] * fx
    if cov_model == "klt":
        # The reference's KLT covariance model (include/features/tracking/pnec_patch.h:128-136,
        # klt_patch_optical_flow.h:377,458): image covariance = top-left 2x2 of the inverse SE(2) Gauss-Newton Hessian of
        # the tracked patch / uncertainty_scaling (10).  H_se2 = sum_i J_i' J_i over the pattern's pixels with
        # J_i = g_i' [I | (-y_i, x_i)'] (image gradient g_i, pattern offset (x_i, y_i)).  Synthetic patches: 52 pixels on
        # rings of radius 1..4 px (the size of the reference's pattern), gradients g_i = Rot(alpha) diag(1, sqrt(rho)) z_i,
        # z_i ~ N(0, I) -- an oriented texture of anisotropy rho ~ U(0.15, 1) (edges to corners) -- in units in which the
        # patch's mean gradient energy is 1; the missing image-noise variance is one global scale (the PNEC weights do not
        # see it), calibrated here so that the tracking noise drawn from these covariances has a std of ~0.15..0.4 px.
        NPAT = 52
        ring = torch.arange(NPAT, **f64)
        rad = 1.0 + 3.0 * (ring % 4) / 3.0
        ang = 2.0 * math.pi * ring / NPAT * 4.0 + 0.4 * (ring % 4)
        px, py = rad * torch.cos(ang), rad * torch.sin(ang)
        alpha = U(M) * math.pi
        rho = 0.15 + 0.85 * U(M)
        cov2d = torch.empty(M, 2, 2, **f64)
        step = 1 << 18
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            zg = torch.randn(m1 - m0, NPAT, 2, generator=g, **f64)
            ca, sa = torch.cos(alpha[m0:m1])[:, None], torch.sin(alpha[m0:m1])[:, None]
            u0, u1 = zg[..., 0], zg[..., 1] * torch.sqrt(rho[m0:m1])[:, None]
            gx, gy = ca * u0 - sa * u1, sa * u0 + ca * u1
            jr = -gx * py + gy * px                                    # d residual / d angle
            J = torch.stack([gx, gy, jr], -1)                          # [m, NPAT, 3]
            H = J.transpose(-1, -2) @ J
            cov2d[m0:m1] = torch.linalg.inv(H)[:, :2, :2] / 10.0 * (0.02 * 10.0 * NPAT)
    else:
        # the simulator's anisotropic-inhomogeneous model (standard_experiments.cc:99-119), ~0.2 px std
        alpha = U(M) * math.pi
        beta = (U(M) + 1.0) / 2.0
        scale = 0.04 * (U(M) + 0.5)
        ca, sa = torch.cos(alpha), torch.sin(alpha)
        rot = torch.stack([ca, -sa, sa, ca], -1).reshape(M, 2, 2)
        dg = torch.zeros(M, 2, 2, **f64)
        dg[:, 0, 0] = beta
        dg[:, 1, 1] = 1 - beta
        cov2d = scale[:, None, None] * (rot @ dg @ rot.transpose(-1, -2))
    cov2d = 0.5 * (cov2d + cov2d.transpose(-1, -2))
    a, b, d = cov2d[:, 0, 0], cov2d[:, 1, 0], cov2d[:, 1, 1]
    l00 = torch.sqrt(a); l10 = b / l00; l11 = torch.sqrt(d - l10 * l10)
    z = torch.randn(M, 2, generator=g, **f64)
    p2n = p2 + torch.stack([l00 * z[:, 0], l10 * z[:, 0] + l11 * z[:, 1], torch.zeros(M, **f64)], -1)
    covs2 = unscented_bearing_cov(p2n, cov2d)
    bvs1 = p1 / torch.linalg.norm(p1, dim=-1, keepdim=True)
    bvs2 = p2n / torch.linalg.norm(p2n, dim=-1, keepdim=True)
    # start: previous relative rotation ~ small perturbation, translation guess = GT + jitter
    axis = _sphere(U(B), U(B))
    R_off = axis_angle_to_matrix(axis, torch.sqrt(U(B)) * 0.005)
    init_R = R_off @ R_gt
    init_t = t_gt / torch.linalg.norm(t_gt, dim=-1, keepdim=True) + 0.01 * (U(B, 3) - 0.5)
    init_t = init_t / torch.linalg.norm(init_t, dim=-1, keepdim=True)
    return (offsets.cpu().numpy(), bvs1, bvs2, covs2, R_gt, t_gt,
            matrix_to_quaternion_xyzw(init_R), init_t)
