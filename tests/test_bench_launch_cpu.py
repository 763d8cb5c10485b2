"""The multi-rank launch path of bench.py, run where there is no GPU: the SAME spawn / rendezvous /
partition / side-stream-free gather code with 2 ranks on gloo and a stubbed solve (--dry-run-cpu).
What a real N-GPU run adds on top is only the device solve and the RCCL backend."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out           # rank 0 prints exactly one JSON line
    return json.loads(lines[0])


def _run(cmd, env=None):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return _line(p.stdout)


def test_plain_invocation_with_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` (no torchrun): must run TWO ranks and say so, not silently one."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "1000",
              "--dry-run-cpu"], env)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["pairs_per_rank"] == [1000, 1000] and d["config"]["pairs_total"] == 2000
    assert d["config"]["launcher"] == "self-spawned ranks"
    assert d["scaling"] == "weak" and d["dry_run"] is True and d["value"] is None
    assert d["records_in_order"] is True   # rank 0 checked order, step stamp and owner of every record


def test_kitti_all_workload_is_partitioned_by_correspondence_count():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "kitti_all",
              "--dry-run-cpu"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["pairs_total"] == 23190 and sum(d["config"]["pairs_per_rank"]) == 23190
    c = d["config"]["corr_per_rank"]
    assert abs(c[0] - c[1]) <= 2 * 700     # balanced to within a couple of pairs' worth of correspondences
    assert d["records_in_order"] is True


def test_torchrun_launch_joins_the_existing_job():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "2",
              "--warmup", "0", "--pairs", "64", "--dry-run-cpu"])
    assert d["n_gpus"] == 2 and d["config"]["launcher"] == "torch.distributed.run"


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-cpu"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_without_gpu_the_real_bench_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)


def test_tracks_format_round_trip_and_sharded_loading(tmp_path):
    from pnec_amd import tracks as tk
    from pnec_amd.distributed import partition
    whole = tk.kitti_all_shard(1000, 1040, frames=(1101, 271))
    tk.validate(whole)
    assert whole.n_pairs == 40 and whole.sequence is not None
    path = str(tmp_path / "t.npz")
    tk.save_tracks(path, whole)
    sizes = tk.sizes_of(path)
    np.testing.assert_array_equal(sizes, whole.sizes)
    b = partition(sizes, 3)
    got = [tk.load_tracks(path, int(b[r]), int(b[r + 1])) for r in range(3)]
    assert sum(g.n_pairs for g in got) == 40
    np.testing.assert_allclose(np.concatenate([g.bvs2 for g in got]), whole.bvs2.numpy())
    np.testing.assert_allclose(np.concatenate([g.init_q for g in got]), whole.init_q.numpy())
    for g in got:
        assert g.offsets[0] == 0 and g.offsets[-1] == len(g.bvs1)
    # any range of the synthetic set is the same whoever builds it (chunk-seeded)
    sub = tk.kitti_all_shard(1010, 1020, frames=(1101, 271))
    o = whole.offsets
    assert bool((sub.bvs1 == whole.bvs1[o[10]:o[20]]).all())
    with pytest.raises(ValueError):
        tk.load_tracks(path, 5, 50)
    bad = tk.Tracks(whole.offsets, whole.bvs1[:-1], whole.bvs2, whole.covs, whole.init_q, whole.init_t)
    with pytest.raises(ValueError, match="bvs1"):
        tk.validate(bad)


def test_kitti_all_set_has_the_real_sequence_lengths():
    from pnec_amd import tracks as tk
    assert tk.kitti_all_num_pairs() == 23190
    s = tk.kitti_all_sizes()
    assert len(s) == 23190 and s.min() >= 64 and 450 < s.mean() < 550
    ids = tk.kitti_all_sequence_ids()
    assert np.bincount(ids).tolist() == [f - 1 for f in tk.KITTI_FRAMES]


def test_profiled_counters_are_dropped_when_the_device_code_changed(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed rocprofv3 record; it must read null unless that
    record was measured on the device code being benchmarked (sha256 of the kernel sources) and on the same
    workload and launch geometry."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    key = ["sim100k", 100000, 512, 10, [8, 1, 3]]
    rec = {"kernel_sources_sha256": bench.kernel_sources_sha256(), "lib_sha256": "0" * 64, "hbm_bytes_per_launch": 4.9e9,
           "valu_busy_frac": 0.88,
           "workload": {"name": "sim100k", "pairs": 100000, "corr": 512, "iters": 10, "geometry": [8, 1, 3]}}
    (tmp_path / "profiles").mkdir()
    (tmp_path / "pnec_amd" / "csrc").mkdir(parents=True)
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"not the library")
    current = rec["kernel_sources_sha256"]
    monkeypatch.setattr(bench, "kernel_sources_sha256", lambda: current)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(rec))
    assert bench.profiled_counters(str(lib), key)[:2] == (4.9e9, 0.88)
    assert bench.profiled_counters(str(lib), ["sim100k", 100000, 512, 10, [8, 2, 3]])[0] is None      # other geometry
    assert bench.profiled_counters(str(lib), ["kitti_all", 23190, 0, 0, [8, 1, 3]])[0] is None         # other workload
    rec["kernel_sources_sha256"] = "f" * 64                                                            # kernels edited since
    (tmp_path / "profiles" / "traffic_latest.json").write_text(json.dumps(rec))
    t, busy, why = bench.profiled_counters(str(lib), key)
    assert t is None and busy is None and "other device code" in why
    (tmp_path / "profiles" / "traffic_latest.json").unlink()
    assert bench.profiled_counters(str(lib), key)[0] is None
    # the committed record belongs to the committed kernels
    monkeypatch.undo()
    committed = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert committed["kernel_sources_sha256"] == bench.kernel_sources_sha256(), \
        "profiles/traffic_latest.json is stale: re-run tools/profile_bench.sh + tools/make_traffic_json.py"


def test_front_flop_model_and_work_counts_belong_to_the_committed_front_stage_code():
    """The chain's stage rooflines multiply committed work counts (profiles/chain_work_latest.json, a -DPNEC_WORK_COUNT run) by
    a flop table in bench.py: both carry the sha256 of the front-stage sources they were made on, a line never uses stale
    counts, and the commit fails here when pnec_frontend.hip moves on without them (the table priced a
    gradient the kernel no longer computed).  The table's straight-line pieces are cross-checked against the compiled code."""
    import json
    import bench
    cur = bench.front_sources_sha256()
    committed = json.load(open(os.path.join(ROOT, "profiles", "chain_work_latest.json")))
    assert committed.get("frontend_sources_sha256") == cur, \
        "profiles/chain_work_latest.json is stale: rebuild the counting library and re-run tools/count_chain_work.py"
    assert bench.FRONT_FLOP_MODEL_STAMP == cur, \
        "bench.py's FLOP_* table was derived on other front-stage code: re-derive it (tools/isa_front_regions.py) and re-stamp"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_front_regions
    r = isa_front_regions.count()
    straight = r["cayley"]["flop"] + r["m"]["flop"] + (r["with_gradient"]["flop"] - r["value_only"]["flop"])
    model = 47 + 339 + 292
    # (the gradient is read as a difference of two compilations: the value-only one also drops the eigenVECTOR's last
    # normalisation and the fallback's vectors, which the model books under the eigenpair -- the difference reads 327..357
    # flop against the model's 292 depending on how the compiler lays the fallback out; cayley and M match to the flop)
    assert r["cayley"]["flop"] <= 51 and r["m"]["flop"] == 339, r
    assert abs(straight - model) <= 0.12 * model, (straight, model, r)
    assert bench.FLOP_ES_POINT == model + 274 and bench.FLOP_ES_QUAD_EVAL == 4 * bench.FLOP_ES_POINT + 160


def test_chain_stage_rooflines_from_committed_work_counts():
    """bench.py --chain prints one roofline block per stage: algorithmic flop = committed work counts
    (profiles/chain_work_latest.json, taken by a -DPNEC_WORK_COUNT build) x the flop model, over the live stage time.
    The counts are keyed by workload (pairs, correspondences): another workload gets blocks without a fraction, never a
    stale one; and the committed file has the two bench workloads with plausible contents."""
    import bench
    c = bench.load_chain_counts("kitti_all_chain", 23190, 11595556)
    assert c is not None and c["ransac_stage"]["ransac_hypotheses"] >= 16 * 23190
    assert 5 < c["ransac_stage"]["ransac_quad_evaluations"] / c["ransac_stage"]["ransac_hypotheses"] < 20
    assert bench.load_chain_counts("kitti_all_chain", 23190, 1) is None and bench.load_chain_counts("nope", 1, 1) is None
    ref = {"stage": "refinement", "bound": "hbm", "frac": 0.3}
    blocks = bench.chain_stage_rooflines(c, {"ransac_es": 2.0, "weighted_es": 1.0}, 10**9, 9 * 10**8, ref)
    assert [b["bound"] for b in blocks] == ["valu_fp64", "valu_fp64", "hbm"] and blocks[2] is ref
    r = blocks[0]
    want = ((c["ransac_stage"]["ransac_quad_evaluations"] + c["ransac_stage"]["es_on_inliers_quad_evaluations"]) * bench.FLOP_ES_QUAD_EVAL
            + c["ransac_stage"]["ransac_scored_tiles"] * 64 * bench.FLOP_SCORE_CORR
            + c["ransac_stage"]["ransac_inlier_pass_corr"] * bench.FLOP_INLIER_CORR
            + c["ransac_stage"]["ransac_hypotheses"] * 10 * bench.FLOP_SAMPLE_CORR)
    assert r["algorithmic_gflop"] == pytest.approx(want / 1e9) and r["achieved"] == pytest.approx(want / 2e-3 / 1e12)
    assert 0.0 < r["frac"] < 1.0 and 0.0 < blocks[1]["frac"] < 1.0          # below the roof at these (realistic) times
    none = bench.chain_stage_rooflines(None, {"ransac_es": 2.0, "weighted_es": 1.0}, 1, 1, ref)
    assert none[0]["frac"] is None and "no committed work counts" in none[0]["note"]


def test_a_rank_that_never_arrives_gives_a_json_error_line_not_a_hang():
    """bench.py --gpus N (N > 1) pre-flight: rank 0 reports the backend, the visible devices, the RCCL version and
    HSA_ENABLE_IPC_MODE_LEGACY on stderr before the first collective, and start-up is bounded: here rank 1 of a 2-rank
    job is never started -- rank 0 must end within its limit with ONE JSON line carrying "error" (value null)."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PNEC_BENCH_JOIN_TIMEOUT_S="6")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-500:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] is None and line["n_gpus"] == 2 and ("init_process_group" in line["error"] or "did not finish" in line["error"])
    pre = json.loads(next(l for l in r.stderr.splitlines() if l.startswith('{"preflight"')))["preflight"]
    assert pre["backend"] == "gloo" and pre["world_size"] == 2 and "HSA_ENABLE_IPC_MODE_LEGACY" in pre


def test_world_size_8_dry_run_of_the_chain_workload_as_the_driver_launches_it():
    """The first real N > 1 run will be the driver's (`python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8 ...`): rehearse its launcher, partition, ragged gather and order check with EIGHT ranks on gloo (stubbed
    solve) for the strong-scaling chain workload -- and the self-diagnosis fields of the line: every rank's own wall
    time and pair count, the digest of the gathered records."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
              "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "8", "--steps", "2",
              "--warmup", "1", "--workload", "kitti_all", "--chain", "--dry-run-cpu"])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["launcher"] == "torch.distributed.run"
    sizes = d["config"]["pairs_per_rank"]
    assert len(sizes) == 8 and sum(sizes) == 23190 and min(sizes) > 2000
    c = d["config"]["corr_per_rank"]
    assert max(c) - min(c) <= 2 * 700          # balanced by correspondence count to within a couple of pairs
    assert d["records_in_order"] is True       # rank 0 checked global order, step stamp and owner of every record
    pr = d["per_rank"]
    assert pr["pairs"] == sizes and len(pr["wall_ms_per_step"]) == 8 and all(x > 0 for x in pr["wall_ms_per_step"])
    assert pr["kernel_ms"] == [None] * 8 and pr["gather_ms"] == [None] * 8     # (no device in a dry run)
    assert len(d["records_sha256"]) == 64
    # the same workload with two ranks gathers the same records (the stub's records depend on the global pair index
    # and the step only): the digest does not depend on the number of ranks, except for the owner column
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    d1 = _run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "kitti_all", "--chain",
               "--dry-run-cpu"], env)
    assert d1["config"]["pairs_per_rank"] == [23190]
