"""Data formats either side of the path: the reference simulator's experiment
CSVs, run_simulation's result tables and the odometry pose stream."""
import json
import os

import numpy as np
import pytest

from pnec_amd import io_formats as io
from pnec_amd import simulation as sim


def test_experiment_folder_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    E, N = 3, 5
    q = rng.normal(size=(E, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    poses_1 = np.concatenate([np.tile([0, 0, 0, 1.0], (E, 1)), np.zeros((E, 3))], 1)
    poses_2 = np.concatenate([q, rng.normal(size=(E, 3))], 1)
    p1 = rng.normal(size=(E, N, 3)) * 100
    p2 = rng.normal(size=(E, N, 3)) * 100
    c1 = -np.tile(np.eye(3), (E, N, 1, 1))
    c2 = np.zeros((E, N, 3, 3)); c2[..., :2, :2] = rng.uniform(0.1, 1, size=(E, N, 1, 1)) * np.eye(2)
    io.write_experiments(tmp_path, poses_1, poses_2, p1, p2, c1, c2)
    # literal format: 7 comma-separated values per pose row, trailing comma on point rows, %g
    first = open(tmp_path / "poses_2.csv").readline().strip().split(",")
    assert len(first) == 7 and first[0] == "%g" % poses_2[0, 0]
    assert open(tmp_path / "points_1.csv").readline().rstrip("\n").endswith(",")
    assert len(open(tmp_path / "covs_2.csv").readline().strip().rstrip(",").split(",")) == 9 * N
    back = io.read_experiments(tmp_path)
    np.testing.assert_allclose(back["poses_2"], poses_2, rtol=1e-5)
    np.testing.assert_allclose(np.stack(back["points_2"]), p2, rtol=1e-5)
    np.testing.assert_allclose(np.stack(back["covs_2"]), c2, rtol=1e-5, atol=1e-12)
    R, t = io.relative_poses(back["poses_1"], back["poses_2"])
    np.testing.assert_allclose(R[1], io.quat_xyzw_to_matrix(poses_2[1, :4]), atol=1e-5)
    np.testing.assert_allclose(t, poses_2[:, 4:], rtol=1e-5)


def test_result_tables_and_pose_stream(tmp_path):
    res = {"PNEC": {"r_error": [0.01, 0.02], "t_error": [1.5, 2.5], "cost": [3.0, 4.0]},
           "NEC": {"r_error": [0.03, 0.04], "t_error": [3.5, 4.5], "cost": [5.0, 6.0]}}
    io.write_result_tables(tmp_path, res)
    lines = open(tmp_path / "r_error.csv").read().splitlines()
    assert lines[0] == "index,PNEC,NEC" and lines[1] == "0,0.01,0.03" and lines[2] == "1,0.02,0.04"
    assert open(tmp_path / "cost.csv").read().splitlines()[2] == "1,4,6"
    g = sim.generate(2, 4, seed=1)
    io.write_pose_file(tmp_path / "poses.txt", [0.5, 1.25], g.R_gt.numpy(), g.t_gt.numpy())
    line = open(tmp_path / "poses.txt").readline().split()
    assert line[0] == "0.500000" and len(line) == 8 and "e" in line[1] and len(line[1].split("e")[0]) in (10, 11)
    ts, t, q = io.read_pose_file(tmp_path / "poses.txt")
    np.testing.assert_allclose(t, g.t_gt.numpy(), rtol=1e-8)
    np.testing.assert_allclose(io.quat_xyzw_to_matrix(q[0]), g.R_gt[0].numpy(), atol=1e-8)
    assert io.TIMING_HEADER.split()[:4] == ["ID", "FrameLoading", "FeatureCreation", "NEC-ES"]


@pytest.mark.gpu
def test_run_simulation_harness_matches_the_oracle_chain(tmp_path, oracle):
    """pnec_amd.run_simulation = the reference's run_simulation ablation (run_simulation.cc:141-180) over a
    simulator experiment folder: CSV in, r_error / t_error / cost tables out, every stage batched on the
    device.  Checked per experiment against the oracle's chain on the same inputs and start poses."""
    import csv
    import torch
    from pnec_amd import io_formats as io
    from pnec_amd import run_simulation as rs
    from pnec_amd import simulation as sim
    E, N = 5, 60
    g = sim.generate(E, N, seed=77)
    rng = np.random.default_rng(4)
    p1 = g.bvs1.numpy() * rng.uniform(2.0, 5.0, size=(E, N, 1))       # 3-D points in frame 1
    p2 = g.bvs2.numpy() / g.bvs2.numpy()[..., 2:3]                       # pinhole points (z = 1) in frame 2
    c2 = np.zeros((E, N, 3, 3))
    s = rng.uniform(0.5, 1.5, size=(E, N)) * (1.0 / 800.0) ** 2
    c2[..., 0, 0], c2[..., 1, 1] = s, 0.6 * s
    q_gt = sim.matrix_to_quaternion_xyzw(g.R_gt).numpy()
    poses_1 = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (E, 1))
    poses_2 = np.concatenate([q_gt, g.t_gt.numpy()], 1)
    folder = str(tmp_path / "exp")
    io.write_experiments(folder, poses_1, poses_2, p1, p2, np.zeros_like(c2), c2)
    assert rs.main([folder, "--seed", "3"]) == 0
    tables = {}
    for name in ("r_error", "t_error", "cost"):
        rows = list(csv.reader(open(os.path.join(folder, name + ".csv"))))
        assert rows[0] == ["index"] + list(rs.METHODS) and len(rows) == E + 1
        tables[name] = np.array([[float(v) for v in r[1:]] for r in rows[1:]])
    assert np.isfinite(tables["r_error"]).all() and (tables["r_error"] < 5.0).all()
    # the same chain through the oracle, from the values the folder actually holds (6 significant digits)
    ex = io.read_experiments(folder)
    R_gt, t_gt = io.relative_poses(ex["poses_1"], ex["poses_2"])
    R0, t0 = rs.perturbed_start(R_gt, t_gt, np.random.default_rng(3), 1.0)
    col = {m: i for i, m in enumerate(rs.METHODS)}
    for e in range(E):
        b1 = ex["points_1"][e] / np.linalg.norm(ex["points_1"][e], axis=1, keepdims=True)
        cov = np.stack([oracle.unscented_transform(p, c) for p, c in zip(ex["points_2"][e], ex["covs_2"][e])])
        b2 = ex["points_2"][e] / np.linalg.norm(ex["points_2"][e], axis=1, keepdims=True)
        Rn, tn = oracle.nec_eigensolver(b1, b2, R0[e])
        Rw, tw = oracle.weighted_eigensolver(b1, b2, cov, Rn, tn, 1e-13, 10)
        sol = oracle.solve(oracle.MODE_TARGET, b1, b2, cov, None, 1e-13, oracle.quat_from_rot(Rw), tw,
                           oracle.default_options())
        assert abs(oracle.rotational_difference_deg(R_gt[e], Rn) - tables["r_error"][e, col["NEC"]]) < 2e-5
        assert abs(oracle.rotational_difference_deg(R_gt[e], sol.R) - tables["r_error"][e, col["PNEC"]]) < 2e-5
        assert tables["r_error"][e, col["PNEC w/o LS"]] == tables["r_error"][e, col["NEC"]]   # the reference's quirk


def test_run_simulation_start_poses_and_metrics():
    """host logic of the run_simulation harness: start-pose perturbation (sim_common.cc:205-231) and the
    error metrics (common.cc:210-235)"""
    from pnec_amd import run_simulation as rs
    rng = np.random.default_rng(0)
    E = 200
    g = sim.generate(E, 8, seed=5)
    R_gt, t_gt = g.R_gt.numpy(), g.t_gt.numpy()
    R0, t0 = rs.perturbed_start(R_gt, t_gt, rng, 1.0)
    assert np.allclose(np.einsum("eij,ekj->eik", R0, R0), np.eye(3), atol=1e-12)        # rotations
    assert np.allclose(np.linalg.norm(t0, axis=1), 1.0, atol=1e-12)                         # unit translations
    ang = np.radians(rs.rotational_difference_deg(R_gt, R0))
    assert ang.max() <= 0.01 + 1e-12 and ang.mean() > 0.004                                # sqrt(u) * 0.01 rad
    # metrics: 90 degree rotation about z; antipodal translations are the same direction
    Rz = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    assert abs(rs.rotational_difference_deg(np.eye(3)[None], Rz[None])[0] - 90.0) < 1e-12
    t = np.array([[0.0, 0, 1]])
    assert rs.translational_difference_deg(t, -t)[0] < 1e-6
    assert abs(rs.translational_difference_deg(t, np.array([[1.0, 0, 0]]))[0] - 90.0) < 1e-12
    assert rs.translational_difference_deg(np.zeros((1, 3)), t)[0] == 90.0                 # degenerate input


# what the reference's `out << timing` (pnec_vo.cc:273-276, timing.cc:49-66) puts into timing.txt for two frames
# with (loading, features, nec-es, it-es, avg-it-es, ceres) = (3, 12, 40, 27, 3, 5) and (0, 0, 1, 0, 0, 2):
# integral millisecond counts, blank separated, OPTIMIZATION = nec + it + ceres, TOTAL = loading + features + that
TIMING_TXT = ("ID FrameLoading FeatureCreation NEC-ES IT-ES AVG-IT-ES CERES OPTIMIZATION TOTAL\n"
              "1 3 12 40 27 3 5 72 87\n"
              "2 0 0 1 0 0 2 3 3\n")


def test_timing_file_is_the_references_table_literally(tmp_path):
    rows = [dict(id=1, frame_loading=3, feature_creation=12, nec_es=40, it_es=27, avg_it_es=3, ceres=5),
            (2, 0, 0, 1, 0, 0, 2)]
    io.write_timing_file(tmp_path / "timing.txt", rows)
    assert open(tmp_path / "timing.txt").read() == TIMING_TXT
    a = io.read_timing_file(tmp_path / "timing.txt")
    assert a.tolist() == [[1, 3, 12, 40, 27, 3, 5, 72, 87], [2, 0, 0, 1, 0, 0, 2, 3, 3]]
    assert io.format_timing_row(7, nec_es=2.9) == "7 0 0 2 0 0 0 2 2"      # duration_cast truncates
    io.write_timing_file(tmp_path / "empty.txt", [])                        # a run without frames: header only
    assert open(tmp_path / "empty.txt").read() == TIMING_TXT.splitlines(True)[0]
    assert io.read_timing_file(tmp_path / "empty.txt").shape == (0, 9)
    (tmp_path / "bad.txt").write_text(TIMING_TXT.replace("72 87", "73 87"))
    with pytest.raises(ValueError):
        io.read_timing_file(tmp_path / "bad.txt")


def test_cpp_facade_timing_streams_the_same_table(tmp_path):
    """pnec::common::FrameTiming / Timing of the host facade (pnec_host.h) stream exactly the reference's
    timing.txt; compiled here against libpnec_host.so (no device call is made)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "pnec_amd")
    if not os.path.exists(os.path.join(pkg, "libpnec_host.so")):
        pytest.skip("libpnec_host.so not built")
    src = tmp_path / "t.cc"
    src.write_text(r'''
#include <iostream>
#include <sstream>
#include "pnec_host.h"
int main(int argc, char **argv) {
  pnec::common::Timing timing;
  pnec::common::FrameTiming a(1);
  a.frame_loading_ = 3; a.feature_creation_ = 12; a.nec_es_ = 40; a.it_es_ = 27; a.avg_it_es_ = 3; a.ceres_ = 5;
  pnec::common::FrameTiming b(2);
  b.nec_es_ = 1; b.ceres_ = 2;
  timing.push_back(a);
  timing.push_back(b);
  std::cout << timing;
  std::ostringstream one;
  one << a;
  if (one.str() != "1 3 12 40 27 3 5 72 87" || a.OptimizationTime() != 72 || a.TotalTime() != 87) return 2;
  return timing.Save(argv[1]) ? 0 : 3;
}
''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + os.path.join(pkg, "csrc", "host"),
                    str(src), "-o", str(exe), "-L" + pkg, "-lpnec_host", "-lpnec_hip", "-Wl,-rpath," + pkg], check=True)
    out = subprocess.run([str(exe), str(tmp_path / "timing.txt")], check=True, capture_output=True, text=True).stdout
    assert out == TIMING_TXT
    assert open(tmp_path / "timing.txt").read() == TIMING_TXT
    assert io.read_timing_file(tmp_path / "timing.txt").shape == (2, 9)


def test_tracks_validator_and_converter_from_the_simulator_folder(tmp_path):
    """pnec_amd/tracks.py (SURVEY 8f row 4 / VERDICT r4 item 8): a simulator folder of the reference
    (experiments.cc:131-172) -> a tracks file through `python -m pnec_amd.tracks from-experiments`; the validator CLI
    accepts it and names what is wrong with broken ones (non-unit bearings, asymmetric / indefinite covariances, rows that
    pair different track ids, pairs below the minimum count); loading the file gives back exactly what was converted."""
    import subprocess
    import sys

    from pnec_amd import simulation as sim
    from pnec_amd import tracks as tk
    rng = np.random.default_rng(3)
    E, N = 5, 40
    poses_1 = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (E, 1))
    poses_2 = poses_1.copy()
    points_1, points_2, covs_1, covs_2 = [], [], [], []
    for e in range(E):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        ang = 0.2 * rng.random()
        poses_2[e, :3], poses_2[e, 3] = ax * np.sin(ang / 2), np.cos(ang / 2)
        poses_2[e, 4:] = rng.normal(size=3) * 0.3
        p1 = np.column_stack([rng.uniform(-300, 300, N), rng.uniform(-200, 200, N), np.full(N, 800.0)])
        p2 = p1 + np.column_stack([rng.normal(size=(N, 2)) * 2.0, np.zeros(N)])
        c = np.zeros((N, 3, 3))
        a = rng.uniform(0.5, 1.5, N); b = rng.uniform(-0.3, 0.3, N); d = rng.uniform(0.5, 1.5, N)
        c[:, 0, 0], c[:, 0, 1], c[:, 1, 0], c[:, 1, 1] = a, b, b, d
        points_1.append(p1); points_2.append(p2); covs_1.append(c.copy()); covs_2.append(c)
    folder = tmp_path / "exp"
    folder.mkdir()
    io.write_experiments(str(folder), poses_1, poses_2, points_1, points_2, covs_1, covs_2)
    out = str(tmp_path / "tracks.npz")
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "pnec_amd.tracks", "from-experiments", str(folder), out], capture_output=True,
                       text=True, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-m", "pnec_amd.tracks", "check", out], capture_output=True, text=True, env=env, cwd=ROOT)
    rep = json.loads(r.stdout)
    assert r.returncode == 0 and rep["ok"] and rep["pairs"] == E and rep["correspondences"] == E * N and rep["matched_ids_equal"]
    tr = tk.load_tracks(out)
    ref = tk.from_experiments(str(folder))
    for k in ("bvs1", "bvs2", "covs", "init_q", "init_t"):
        np.testing.assert_array_equal(np.asarray(getattr(tr, k)), np.asarray(getattr(ref, k)))
    assert tk.sizes_of(out).tolist() == [N] * E
    # the frame-2 covariance is the reference's UnscentedTransform of the 2x2 image covariance (pinned by goldens elsewhere)
    import torch
    back = io.read_experiments(str(folder))     # (the CSVs hold %g values: the file is the data)
    want = sim.unscented_bearing_cov(torch.from_numpy(back["points_2"][0]), torch.from_numpy(back["covs_2"][0][:, :2, :2])).numpy()
    np.testing.assert_array_equal(np.asarray(tr.covs)[:N], want)
    # broken files: every defect is named
    def broken(**kw):
        t = tk.load_tracks(out)
        for k, f in kw.items():
            setattr(t, k, f(np.array(getattr(t, k))))
        return tk.check(t, min_corr=10)
    assert any("not unit length" in p for p in broken(bvs1=lambda a: a * 1.001)["problems"])
    def asym(c):
        c[:, 0, 1] += 1e-6 * np.abs(c).max(); return c
    assert any("not symmetric" in p for p in broken(covs=asym)["problems"])
    assert any("positive semi-definite" in p for p in broken(covs=lambda c: -c)["problems"])
    assert any("different track ids" in p for p in broken(ids2=lambda i: np.roll(i, 1))["problems"])
    assert any("fewer than" in p for p in tk.check(tk.load_tracks(out), min_corr=N + 1)["problems"])
    bad = str(tmp_path / "bad.npz")
    np.savez(bad, offsets=np.array([0, 3]), bvs1=np.zeros((2, 3)))
    r = subprocess.run([sys.executable, "-m", "pnec_amd.tracks", "check", bad], capture_output=True, text=True, env=env, cwd=ROOT)
    assert r.returncode == 1 and not json.loads(r.stdout)["ok"]


def test_kitti_like_covariances_follow_the_klt_patch_model():
    """generate_kitti_like draws its 2x2 image covariances from the reference's KLT model (top-left block of the inverse
    SE(2) patch Hessian / 10: pnec_patch.h:128-136, klt_patch_optical_flow.h:377,458) since round 5: symmetric positive
    definite, anisotropic (oriented texture), tracking noise of a fraction of a pixel; the simulator's model stays
    selectable and the set is a function of the seed."""
    from pnec_amd import simulation as sim
    from pnec_amd import tracks as tk
    a = sim.generate_kitti_like(6, 200, seed=4)
    b = sim.generate_kitti_like(6, 200, seed=4)
    c = sim.generate_kitti_like(6, 200, seed=4, cov_model="simulator")
    assert np.array_equal(a[3].numpy(), b[3].numpy()) and not np.array_equal(a[3].numpy(), c[3].numpy())
    rep = tk.check(tk.Tracks(a[0], a[1], a[2], a[3], a[6], a[7]))
    assert rep["ok"], rep["problems"]
    w = np.linalg.eigvalsh(a[3].numpy())
    px_std = np.sqrt(w[:, 2]) * 718.856
    assert 0.03 < np.median(px_std) < 0.5 and np.median(w[:, 2] / w[:, 1]) > 1.3     # anisotropic, sub-pixel


@pytest.mark.gpu
def test_tracks_file_from_the_converter_solves_like_the_in_memory_arrays(tmp_path):
    """a tracks file written by the converter, loaded back (whole and as two shards), gives the refinement and the whole
    chain exactly the bits the in-memory arrays give; and `bench.py --workload kitti_all --tracks` runs on it"""
    import subprocess
    import sys

    from pnec_amd import Batch, capi
    from pnec_amd import tracks as tk
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = sim.generate(12, 120, seed=8)
    E, N = 12, 120
    poses_1 = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], dtype=np.float64), (E, 1))
    poses_2 = np.concatenate([g.q_gt.numpy() if hasattr(g, "q_gt") else sim.matrix_to_quaternion_xyzw(g.R_gt).numpy(), g.t_gt.numpy()], 1)
    rng = np.random.default_rng(1)
    p1 = [np.column_stack([rng.uniform(-300, 300, N), rng.uniform(-200, 200, N), np.full(N, 800.0)]) for _ in range(E)]
    # frame-2 image points of the same landmarks under the pose (depth 5): what the simulator writes
    p2, c2 = [], []
    for e in range(E):
        R, t = io.quat_xyzw_to_matrix(poses_2[e, :4]), poses_2[e, 4:]
        X = p1[e] / 800.0 * 5.0
        Y = (R.T @ (X - t).T).T
        p2.append(np.column_stack([Y[:, 0] / Y[:, 2] * 800.0 + rng.normal(size=N) * 0.5, Y[:, 1] / Y[:, 2] * 800.0 + rng.normal(size=N) * 0.5, np.full(N, 800.0)]))
        c = np.zeros((N, 3, 3)); c[:, 0, 0] = 0.3; c[:, 1, 1] = 0.2; c[:, 0, 1] = c[:, 1, 0] = 0.05
        c2.append(c)
    folder = tmp_path / "exp"; folder.mkdir()
    io.write_experiments(str(folder), poses_1, poses_2, p1, p2, c2, c2)
    mem = tk.from_experiments(str(folder))
    path = str(tmp_path / "t.npz")
    tk.save_tracks(path, mem)
    assert tk.check(tk.load_tracks(path))["ok"]

    def solve(tr):
        with Batch(capi.MODE_TARGET, tr.offsets) as b:
            b.fill(np.asarray(tr.bvs1), np.asarray(tr.bvs2), np.asarray(tr.covs))
            r = b.solve(np.asarray(tr.init_q), np.asarray(tr.init_t))
            q, t = b.solve_pipeline(np.asarray(tr.init_q), np.asarray(tr.init_t))
            return np.asarray(r.q), np.asarray(r.t), np.asarray(q), np.asarray(t)
    a = solve(mem)
    b = solve(tk.load_tracks(path))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    s0, s1 = solve(tk.load_tracks(path, 0, 5)), solve(tk.load_tracks(path, 5, 12))
    np.testing.assert_array_equal(np.concatenate([s0[0], s1[0]]), a[0])       # the refinement does not depend on the sharding
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "kitti_all", "--tracks", path, "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["data"].startswith("tracks:") and line["value"] > 0
