"""On-disk formats either side of the hot path, so the reference's
evaluation scripts can consume this build's outputs and this build can ingest the reference
simulator's experiment folders.

  simulator experiments   BaseExperiments::SaveExperiments  src/simulation/experiments.cc:131-172
                          ReadExperiments                   src/simulation/sim_common.cc:109-236
      poses_{1,2}.csv   one row per experiment: qx,qy,qz,qw,tx,ty,tz
      points_{1,2}.csv  one row per experiment: x,y,z, repeated N times (trailing comma)
      covs_{1,2}.csv    one row per experiment: 9 row-major values per point (trailing comma)
      numbers use the C++ stream default precision (6 significant digits, Eigen::StreamPrecision)
  result tables           run_simulation                    src/run_simulation.cc:348-391
      r_error.csv / t_error.csv / cost.csv: header "index,<method>,..." then one row per experiment
  pose stream             pnec::out::SavePose               src/io/odometry_output.cc:43-54
      "<timestamp fixed> tx ty tz qx qy qz qw" with 8-digit scientific notation
  timing                  FrameTiming / Timing operator<<   include/common/timing.h:52-55, src/common/timing.cc:49-67
      timing.txt (pnec_vo.cc:273-276): the header line, then per frame
      "id loading features nec-es it-es avg-it-es ceres optimization total" -- integral milliseconds
      (the reference sets std::scientific on the stream, but every field is an integer count)
"""
from __future__ import annotations

import os

import numpy as np

TIMING_HEADER = "ID FrameLoading FeatureCreation NEC-ES IT-ES AVG-IT-ES CERES OPTIMIZATION TOTAL"


def _g6(x: float) -> str:
    """C++ ostream default formatting: %g with 6 significant digits."""
    return "%g" % x


def write_experiments(folder, poses_1, poses_2, points_1, points_2, covs_1, covs_2) -> None:
    """poses_*: [E,7] (qx,qy,qz,qw,tx,ty,tz); points_*: [E,N,3]; covs_*: [E,N,3,3]."""
    os.makedirs(folder, exist_ok=True)
    for name, arr in (("poses_1", poses_1), ("poses_2", poses_2)):
        with open(os.path.join(folder, name + ".csv"), "w") as f:
            for row in np.asarray(arr, dtype=np.float64):
                f.write(",".join(_g6(v) for v in row) + "\n")
    for name, arr in (("points_1", points_1), ("points_2", points_2)):
        with open(os.path.join(folder, name + ".csv"), "w") as f:
            for exp in np.asarray(arr, dtype=np.float64):
                f.write("".join(",".join(_g6(v) for v in p) + "," for p in exp) + "\n")
    for name, arr in (("covs_1", covs_1), ("covs_2", covs_2)):
        with open(os.path.join(folder, name + ".csv"), "w") as f:
            for exp in np.asarray(arr, dtype=np.float64):
                f.write("".join(",".join(_g6(v) for v in c.reshape(9)) + "," for c in exp) + "\n")


def _rows(path):
    with open(path) as f:
        for line in f:
            cells = [c for c in line.strip().split(",") if c != ""]
            if cells:
                yield np.array([float(c) for c in cells])


def read_experiments(folder):
    """-> dict(poses_1 [E,7], poses_2 [E,7], points_1 [E][N,3], points_2, covs_2 [E][N,3,3]).
    covs_1 is read with the reference's stride-3 quirk left out (it is unused downstream,
    SURVEY Appendix C10): the file's values are returned as written."""
    out = {}
    for name in ("poses_1", "poses_2"):
        out[name] = np.stack(list(_rows(os.path.join(folder, name + ".csv"))))
    for name in ("points_1", "points_2"):
        out[name] = [r.reshape(-1, 3) for r in _rows(os.path.join(folder, name + ".csv"))]
    for name in ("covs_1", "covs_2"):
        p = os.path.join(folder, name + ".csv")
        if os.path.exists(p):
            out[name] = [r.reshape(-1, 3, 3) for r in _rows(p)]
    return out


def quat_xyzw_to_matrix(q):
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def relative_poses(poses_1, poses_2):
    """rel = pose_1^-1 * pose_2 (sim_common.cc:133-135) -> (R [E,3,3], t [E,3])"""
    Rs, ts = [], []
    for a, b in zip(poses_1, poses_2):
        R1, R2 = quat_xyzw_to_matrix(a[:4]), quat_xyzw_to_matrix(b[:4])
        Rs.append(R1.T @ R2)
        ts.append(R1.T @ (b[4:7] - a[4:7]))
    return np.stack(Rs), np.stack(ts)


def write_result_tables(folder, results: dict) -> None:
    """results: {method_name: {"r_error": [E], "t_error": [E], "cost": [E]}} -> the three CSVs of
    run_simulation.cc:348-391 (degrees, degrees, CostFunction)."""
    os.makedirs(folder, exist_ok=True)
    names = list(results)
    n = len(next(iter(results.values()))["r_error"])
    for key, fname in (("r_error", "r_error.csv"), ("t_error", "t_error.csv"), ("cost", "cost.csv")):
        with open(os.path.join(folder, fname), "w") as f:
            f.write("index" + "".join("," + m for m in names) + "\n")
            for i in range(n):
                f.write(str(i) + "".join("," + _g6(float(results[m][key][i])) for m in names) + "\n")


def format_pose_line(timestamp: float, R, t) -> str:
    """SavePose: fixed timestamp, then translation and quaternion (x,y,z,w of Quaterniond(R)) in
    8-digit scientific notation."""
    from .simulation import matrix_to_quaternion_xyzw
    import torch
    q = matrix_to_quaternion_xyzw(torch.as_tensor(np.asarray(R, dtype=np.float64))[None])[0].numpy()
    vals = list(np.asarray(t, dtype=np.float64)) + list(q)
    return "%f" % timestamp + "".join(" %.8e" % v for v in vals)


def write_pose_file(path, timestamps, Rs, ts, append: bool = False) -> None:
    with open(path, "a" if append else "w") as f:
        for tm, R, t in zip(timestamps, Rs, ts):
            f.write(format_pose_line(tm, R, t) + "\n")


def read_pose_file(path):
    """-> (timestamps [M], t [M,3], q_xyzw [M,4])"""
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


TIMING_FIELDS = ("id", "frame_loading", "feature_creation", "nec_es", "it_es", "avg_it_es", "ceres")


def format_timing_row(id, frame_loading=0, feature_creation=0, nec_es=0, it_es=0, avg_it_es=0, ceres=0) -> str:
    """FrameTiming's operator<< (timing.cc:49-58): the six stage times in whole milliseconds, then
    OptimizationTime() = nec_es + it_es + ceres and TotalTime() = loading + features + optimisation
    (timing.cc:40-47).  Durations are truncated to whole milliseconds like duration_cast does."""
    v = [int(id)] + [int(x) for x in (frame_loading, feature_creation, nec_es, it_es, avg_it_es, ceres)]
    optimization = v[3] + v[4] + v[6]
    return " ".join(str(x) for x in v + [optimization, v[1] + v[2] + optimization])


def write_timing_file(path, rows) -> None:
    """Timing's operator<< (timing.cc:60-66) into a truncated file (pnec_vo.cc:273-276).
    rows: dicts with TIMING_FIELDS keys (missing stage times are 0) or 7-sequences in that order."""
    with open(path, "w") as f:
        f.write(TIMING_HEADER + "\n")
        for r in rows:
            f.write((format_timing_row(**r) if isinstance(r, dict) else format_timing_row(*r)) + "\n")


def read_timing_file(path):
    """-> int64 [M,9] (the columns of TIMING_HEADER); checks the header and the two derived columns."""
    with open(path) as f:
        header = f.readline().rstrip("\n")
        if header != TIMING_HEADER:
            raise ValueError(f"not a timing file: header {header!r}")
        a = np.array([[int(c) for c in line.split()] for line in f if line.strip()], dtype=np.int64).reshape(-1, 9)
    if len(a) and not (np.array_equal(a[:, 7], a[:, 3] + a[:, 4] + a[:, 6])
                       and np.array_equal(a[:, 8], a[:, 1] + a[:, 2] + a[:, 7])):
        raise ValueError("timing rows whose OPTIMIZATION / TOTAL columns are not the sums of their stages")
    return a
