"""Two ranks, the REAL solver: every rank solves its shard of one ragged batch on the GPU (both on cuda:0 -- the
test box has one; the collective runs on gloo with the records staged through host memory), the result records
are gathered on rank 0 with pnec_amd.distributed.gather_records and must equal the one-process solve of the
whole batch bit for bit, in order.  (The RCCL path itself needs two GPUs: bench.py --gpus N.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["PNEC_ROOT"])
import numpy as np, torch, torch.distributed as dist
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd.distributed import partition, pack_records, gather_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
P = 96
offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=300, seed=17)     # ragged sizes
off = np.asarray(offsets)
bounds = partition(np.diff(off), world)
a, b = int(bounds[rank]), int(bounds[rank + 1])
sl = slice(off[a], off[b])
dev = torch.device("cuda:0")
with Batch(capi.MODE_TARGET, off[a:b + 1] - off[a], device=0) as batch:
    batch.fill(f1[sl].to(dev), f2[sl].to(dev), c2[sl].to(dev))
    res = batch.solve(q0[a:b].to(dev), t0[a:b].to(dev))
    rec = pack_records(res).cpu()
sizes = [int(bounds[r + 1] - bounds[r]) for r in range(world)]
got = gather_records(rec, world, rank, sizes=sizes, dst=0)
if rank == 0:
    with Batch(capi.MODE_TARGET, off, device=0) as batch:
        batch.fill(f1.to(dev), f2.to(dev), c2.to(dev))
        ref = pack_records(batch.solve(q0.to(dev), t0.to(dev))).cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert min(sizes) > 0
    print("GATHERED_EQUALS_WHOLE_BATCH", sizes)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_solve_their_shards_and_gather_the_whole_batch():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PNEC_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "GATHERED_EQUALS_WHOLE_BATCH" in outs[0][0]


def test_config5_full_set_as_eight_shards_equals_one_batch_and_the_oracle(oracle):
    """BASELINE config 5 at its size: all 23 190 ragged pairs of the KITTI-00..10-sized set (the labelled synthetic
    stand-in: no KITTI data exists here), (a) solved as ONE batch, (b) solved as the 8 contiguous, correspondence-
    balanced shards `partition` hands to 8 ranks -- each shard built from scratch the way its rank would build it
    (tracks.kitti_all_shard(a, b)) -- with the records concatenated in rank order, which is what the one gather
    delivers (scripts/parallel_kitti.sh:60-69 is the reference's fan-out).  The two must be equal bit for bit,
    and 256 sampled pairs must match the reference-faithful oracle (central differences, Ceres-default
    termination): iteration counts equal, rotations <= 1e-6 rad."""
    import numpy as np
    import torch
    from pnec_amd import Batch, capi
    from pnec_amd import tracks as tk
    from pnec_amd.distributed import pack_records, partition
    dev = torch.device("cuda:0")
    sizes = tk.kitti_all_sizes()
    P = len(sizes)
    assert P == 23190 and sizes.min() >= 64 and sizes.max() > 512          # ragged, beyond one wavefront's 512
    whole = tk.kitti_all_shard(0, P, device=dev)
    assert np.array_equal(np.diff(whole.offsets), sizes)
    with Batch(capi.MODE_TARGET, whole.offsets) as b:
        b.fill(whole.bvs1, whole.bvs2, whole.covs)
        ref = pack_records(b.solve(whole.init_q.contiguous(), whole.init_t.contiguous())).cpu()
    bounds = partition(sizes, 8)
    shard_corr = [int(sizes[bounds[r]:bounds[r + 1]].sum()) for r in range(8)]
    assert max(shard_corr) - min(shard_corr) <= 2 * sizes.max()            # balanced by correspondences
    parts = []
    for r in range(8):
        a, c = int(bounds[r]), int(bounds[r + 1])
        tr = tk.kitti_all_shard(a, c, device=dev)                           # what rank r builds for itself
        with Batch(capi.MODE_TARGET, tr.offsets) as b:
            b.fill(tr.bvs1, tr.bvs2, tr.covs)
            parts.append(pack_records(b.solve(tr.init_q.contiguous(), tr.init_t.contiguous())).cpu())
    got = torch.cat(parts)
    assert got.shape == ref.shape == (P, 10)
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # sampled parity against the oracle (the reference's execution: numeric central differences)
    rng = np.random.default_rng(5)
    pick = np.sort(rng.choice(P, 256, replace=False))
    off = whole.offsets
    idx = np.concatenate([np.arange(off[p], off[p + 1]) for p in pick])
    f1, f2, cv = (x[torch.as_tensor(idx, device=dev)].cpu().numpy() for x in (whole.bvs1, whole.bvs2, whole.covs))
    soff = np.concatenate([[0], np.cumsum(sizes[pick])])
    q, t, cost, its, st = oracle.solve_batch(oracle.MODE_TARGET, soff, f1, f2, oracle.covs_to_colmajor9(cv), None, 1e-13,
                                             whole.init_q[pick].cpu().numpy(), whole.init_t[pick].cpu().numpy(),
                                             options=oracle.default_options(jacobian_mode=oracle.JAC_NUMERIC_CENTRAL))
    rec = ref.numpy()[pick]
    dots = np.abs(np.sum(rec[:, :4] * q, axis=1)).clip(0, 1)
    ang = 2.0 * np.arccos(dots)
    # arccos loses half the digits near 1: recompute small angles from the vector part
    v = np.stack([rec[:, 3] * q[:, 0] - rec[:, 0] * q[:, 3] - rec[:, 1] * q[:, 2] + rec[:, 2] * q[:, 1],
                  rec[:, 3] * q[:, 1] + rec[:, 0] * q[:, 2] - rec[:, 1] * q[:, 3] - rec[:, 2] * q[:, 0],
                  rec[:, 3] * q[:, 2] - rec[:, 0] * q[:, 1] + rec[:, 1] * q[:, 0] - rec[:, 2] * q[:, 3]], 1)
    ang = 2.0 * np.arctan2(np.linalg.norm(v, axis=1), dots)
    assert ang.max() <= 1e-6, ang.max()
    assert (rec[:, 8].astype(np.int64) == its).mean() >= 0.99, (rec[:, 8], its)   # a stopping test inside rounding: <= 2 of 256
    assert np.abs(rec[:, 8] - its).max() <= 1


def _run_bench(args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_share_the_gpu_config5_device_side_gather():
    """`bench.py --gpus 2 --share-gpu --workload kitti_all`: two ranks, both on cuda:0, the REAL solver on each
    rank's `partition` shard of the 23 190-pair set, RecordGather's device form with world > 1 (solve-done event,
    pack + copy on the side stream, double buffering, per-rank `sizes=` trimming; gloo carries the pinned host
    records because RCCL refuses two ranks on one device).  bench.py itself asserts the gathered shape and
    finiteness; here: the line, the sizes, and that the gathered records equal a one-rank run's bit for bit."""
    two = _run_bench(["--gpus", "2", "--share-gpu", "--workload", "kitti_all", "--steps", "3", "--warmup", "1",
                      "--no-cpu-baseline"])
    assert two["shared_gpu"]["ranks"] == 2 and two["n_gpus"] == 1 and two["scaling"] == "strong"
    assert two["config"]["pairs_total"] == 23190 and sum(two["config"]["pairs_per_rank"]) == 23190
    assert len(two["config"]["pairs_per_rank"]) == 2 and min(two["config"]["pairs_per_rank"]) > 10000
    assert two["shared_gpu"]["collectives_issued"] == 3 + 1 + 1              # steps + warm-up + communicator set-up step
    assert two["value"] > 0 and "gloo" in two["config"]["sharding"]
    it2 = two["config"]["lm_iterations_done_min_mean_max"]
    one = _run_bench(["--gpus", "1", "--workload", "kitti_all", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert one["n_gpus"] == 1 and one["config"]["pairs_total"] == 23190
    # rank 0's shard is the first half: its iteration statistics are a subset's; the whole-set check is the digest
    assert two["records_sha256"] == one["records_sha256"], (two["records_sha256"], one["records_sha256"])
    assert it2[0] >= one["config"]["lm_iterations_done_min_mean_max"][0]


def test_bench_chain_workload_two_ranks_share_the_gpu():
    """config 5 with something to scale: the whole PNEC::Solve chain per pair (`--chain`), two ranks on one GPU
    against one rank: same gathered records"""
    two = _run_bench(["--gpus", "2", "--share-gpu", "--workload", "kitti_all", "--chain", "--steps", "2", "--warmup", "1"])
    one = _run_bench(["--gpus", "1", "--workload", "kitti_all", "--chain", "--steps", "2", "--warmup", "1"])
    assert two["unit"] == one["unit"] == "pairs/s" and "chain" in two
    # one roofline block per stage (round 4); the work counts behind the flop figures belong to the one-rank synthetic set
    assert [b["bound"] for b in one["roofline"]] == ["valu_fp64", "valu_fp64", "hbm"] and one["roofline"][0]["frac"] is not None
    assert two["roofline"][0]["frac"] is None and two["roofline"][2]["frac"] > 0
    assert 0.8 < two["chain"]["inlier_share_mean"] < 0.95                    # 10 % gross mismatches rejected
    assert two["records_sha256"] == one["records_sha256"]
    # ... and steps in flight (each on its own stream and copy of the batch; the default is three) against one at a time
    assert one["chain"]["steps_in_flight"] == 3 and one["chain"]["pairs_per_s_one_step_at_a_time"] > 0
    seq = _run_bench(["--gpus", "1", "--workload", "kitti_all", "--chain", "--steps", "2", "--warmup", "1", "--in-flight", "1"])
    assert seq["chain"]["steps_in_flight"] == 1 and seq["records_sha256"] == one["records_sha256"]


RCCL_ONE_RANK = r'''
import os, sys
sys.path.insert(0, os.environ["PNEC_ROOT"])
import torch, torch.distributed as dist
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd.distributed import RecordGather, pack_records
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # backend "nccl" IS RCCL on ROCm
g = sim.generate(300, 200, seed=9, device=dev)
gather = RecordGather(1, 0, sizes=[300], device=dev, force_collective=True)
outs = [None, None]
with Batch.uniform(capi.MODE_TARGET, 300, 200) as b:
    b.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3))
    for step in range(5):
        slot = gather.acquire()
        outs[slot] = b.solve(g.init_q, g.init_t, out=outs[slot])
        gather.submit(slot, outs[slot])
    got = gather.drain()
    torch.cuda.synchronize()
    want = pack_records(b.solve(g.init_q, g.init_t))
assert gather.collectives == 5
assert got.is_cuda and torch.equal(got, want)
print("RCCL_GATHER_ON_SIDE_STREAM_OK")
dist.destroy_process_group()
'''


def test_rccl_gather_of_device_records_on_the_side_stream_one_rank_communicator():
    """The production form of RecordGather -- dist.gather on DEVICE tensors through backend "nccl" (= RCCL),
    enqueued on the side stream behind the solve-done event -- on the one GPU this box has: a one-rank RCCL
    communicator with the collective forced.  What it cannot show is a second device; what it does show is the
    RCCL call path, stream ordering and buffer lifetime of the code the multi-GPU run executes."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PNEC_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", RCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    assert "RCCL_GATHER_ON_SIDE_STREAM_OK" in r.stdout


def test_one_process_multi_device_entry_equals_the_single_device_call():
    """pnec_hip_solve_pipeline_multi / PNEC::SolveBatch(pairs, devices) (round 4): the batch sharded over a device LIST
    from one process -- contiguous ranges by pnec_hip_partition, one host thread + batch + stream per entry, RANSAC draws
    by global pair index.  With the one GPU a box has: devices = [0, 0] and [0, 0, 0] (two / three handles and threads
    on the same device) must give the single-device call's poses, masks and counts bit for bit, ragged sizes included."""
    import numpy as np
    from pnec_amd import Batch, capi
    from pnec_amd import simulation as sim
    rng = np.random.default_rng(12)
    P = 700
    counts = rng.integers(40, 600, size=P).astype(np.int64)
    counts[:3] = [5, 513, 64]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    g = sim.generate(1, int(off[-1]), seed=5)
    poses = sim.generate(P, 4, seed=6)
    f1, f2, cv = g.bvs1[0].numpy(), g.bvs2[0].numpy().copy(), g.covs2[0].numpy()
    bad = rng.random(len(f2)) < 0.1
    v = rng.normal(size=(int(bad.sum()), 3))
    f2[bad] = v / np.linalg.norm(v, axis=1, keepdims=True)
    c9 = np.ascontiguousarray(np.transpose(cv, (0, 2, 1)).reshape(-1, 9))
    q0, t0 = poses.init_q.numpy().copy(), poses.init_t.numpy().copy()
    L = capi.lib()
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, cv)
        q, t, mask, cnt = b.solve_pipeline(q0, t0, want_inliers=True)
    for devs in ([0], [0, 0], [0, 0, 0]):
        d = np.asarray(devs, dtype=np.int32)
        oq, ot = np.zeros((P, 4)), np.zeros((P, 3))
        om, oc = np.zeros(int(off[-1]), dtype=np.uint8), np.zeros(P, dtype=np.int32)
        capi.check(L.pnec_hip_solve_pipeline_multi(len(d), d.ctypes.data, P, off.ctypes.data, f1.ctypes.data, f2.ctypes.data,
                                                   c9.ctypes.data, q0.ctypes.data, t0.ctypes.data, None, oq.ctypes.data,
                                                   ot.ctypes.data, om.ctypes.data, oc.ctypes.data))
        np.testing.assert_array_equal(oq, np.asarray(q)), devs
        np.testing.assert_array_equal(ot, np.asarray(t))
        np.testing.assert_array_equal(om, np.asarray(mask).astype(np.uint8))
        np.testing.assert_array_equal(oc, np.asarray(cnt))
    # a device that does not exist is an error, not a hang
    d = np.asarray([0, 99], dtype=np.int32)
    assert L.pnec_hip_solve_pipeline_multi(2, d.ctypes.data, P, off.ctypes.data, f1.ctypes.data, f2.ctypes.data, c9.ctypes.data,
                                           q0.ctypes.data, t0.ctypes.data, None, oq.ctypes.data, ot.ctypes.data, None, None) == -1
    # the facade's overload (C++ through pybind): the same poses and inlier lists as its single-device SolveBatch
    import pypnec
    lists = lambda a: [a[off[p]:off[p + 1]] for p in range(40)]
    T0 = [np.vstack([np.hstack([poses.init_R[p].numpy(), poses.init_t[p].numpy()[:, None]]), [0, 0, 0, 1]]) for p in range(40)]
    one, inl_one = pypnec.solve_batch(lists(f1), lists(f2), lists(cv), T0)
    two, inl_two = pypnec.solve_batch(lists(f1), lists(f2), lists(cv), T0, devices=[0, 0])
    for a, bb in zip(one, two):
        np.testing.assert_array_equal(a, bb)
    assert list(inl_one) == list(inl_two)


def test_default_bench_line_carries_the_other_configs_and_the_chain():
    """`python bench.py` on one GPU (what the driver runs): besides the headline the ONE JSON line has a `secondary` array
    -- BASELINE configs 3, 4, 5 and the whole PNEC::Solve chain, measured in the same process -- each with value, unit,
    ms_per_step, roofline and parity, none of them failed, parity within the north star's tolerance, the chain entry with
    one roofline block per stage (run here with fewer steps and a shorter headline batch)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--pairs", "20000",
                        "--quick-secondary", "--cpu-sample", "64"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["metric"].startswith("PNEC pose solves/sec") and line["value"] > 1e6 and "roofline" in line and "cpu_baseline" in line
    sec = line["secondary"]
    assert len(sec) == 6 and not any("error" in e for e in sec), [e.get("error") for e in sec]
    for e in sec:
        assert e["value"] > 0 and e["unit"] in ("solves/s", "pairs/s") and e["ms_per_step"] > 0 and "roofline" in e
        p = e["parity"]
        assert p.get("max_rot_err_rad", 0.0) <= 1e-6 and p.get("bitwise_equal_to_the_batched_call", True) and p.get("bitwise_equal_to_one_call_per_frame", True)
    chain = sec[1]
    assert chain["parity"]["inlier_masks_identical"] and [b["bound"] for b in chain["roofline"]] == ["valu_fp64", "valu_fp64", "hbm"]
    assert all(b["frac"] is not None and 0 < b["frac"] < 1 for b in chain["roofline"])
    assert chain["value"] >= chain["pairs_per_s_one_call_at_a_time"] * 0.9


def test_persistent_multi_device_handle_is_bitwise_the_single_device_calls_and_allocates_nothing(oracle):
    """pnec_hip_multi_* (ABI 5): the refinement -- what north_star shards -- and the whole chain through the persistent
    handle on device lists [0, 0] and [0, 0, 0] (one GPU here: every entry is cuda:0, each with its own batch, stream and
    host thread): bit-identical to the single-device calls, multi-hypothesis starts included; and after the first calls a
    loop of fill + solve + solve_pipeline makes no hipMalloc (pnec_hip_alloc_counters)."""
    import numpy as np
    from pnec_amd import Batch, capi
    from pnec_amd import simulation as sim
    from pnec_amd.multi import MultiBatch, alloc_counters
    rng = np.random.default_rng(3)
    sizes = rng.integers(64, 600, size=96)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    g = sim.generate(len(sizes), int(sizes.max()), seed=21)
    f1 = np.concatenate([g.bvs1[p].numpy()[:n] for p, n in enumerate(sizes)])
    f2 = np.concatenate([g.bvs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    cv = np.concatenate([g.covs2[p].numpy()[:n] for p, n in enumerate(sizes)])
    bad = rng.random(len(f2)) < 0.1
    junk = rng.normal(size=(len(f2), 3))
    f2[bad] = (junk / np.linalg.norm(junk, axis=1, keepdims=True))[bad]
    q0, t0 = g.init_q.numpy(), g.init_t.numpy()
    H = 4
    hyp = rng.normal(size=(len(sizes) * H, 3)); hyp /= np.linalg.norm(hyp, axis=1, keepdims=True)
    with Batch(capi.MODE_TARGET, off) as b:
        b.fill(f1, f2, cv)
        ref = b.solve(q0, t0)
        ref_h = b.solve(q0, None, hyp_t=hyp, n_hyp=H)
        rq, rt, rm, rc = b.solve_pipeline(q0, t0, want_inliers=True)
    for devs in ([0, 0], [0, 0, 0]):
        with MultiBatch(devs, capi.MODE_TARGET, len(sizes), int(off[-1]), int(sizes.max())) as mb:
            mb.fill(off, f1, f2, cv)
            bnd = mb.bounds
            assert bnd[0] == 0 and bnd[-1] == len(sizes) and (np.diff(bnd) > 0).all()
            r = mb.solve(q0, t0)
            np.testing.assert_array_equal(r["q"], np.asarray(ref.q)); np.testing.assert_array_equal(r["t"], np.asarray(ref.t))
            np.testing.assert_array_equal(r["iterations"], np.asarray(ref.iterations))
            np.testing.assert_array_equal(r["status"], np.asarray(ref.status))
            rh = mb.solve(q0, None, hyp_t=hyp, n_hyp=H)
            np.testing.assert_array_equal(rh["q"], np.asarray(ref_h.q)); np.testing.assert_array_equal(rh["cost"], np.asarray(ref_h.cost))
            q, t, m, c = mb.solve_pipeline(q0, t0, want_inliers=True)
            np.testing.assert_array_equal(q, np.asarray(rq)); np.testing.assert_array_equal(t, np.asarray(rt))
            np.testing.assert_array_equal(m, np.asarray(rm)); np.testing.assert_array_equal(c, np.asarray(rc))
            k = 40
            small = (off[:k + 1], f1[:off[k]], f2[:off[k]], cv[:off[k]])
            mb.fill(*small); mb.solve(q0[:k], t0[:k]); mb.solve_pipeline(q0[:k], t0[:k])   # (the second shape's first calls)
            a0 = alloc_counters()
            for _ in range(4):                       # both shapes again, in the same handle
                mb.fill(off, f1, f2, cv)
                mb.solve(q0, t0)
                mb.solve_pipeline(q0, t0)
                mb.fill(*small)
                r2 = mb.solve(q0[:k], t0[:k])
                mb.solve_pipeline(q0[:k], t0[:k])
            np.testing.assert_array_equal(r2["q"], np.asarray(ref.q)[:k])
            a1 = alloc_counters()
            assert a1["hip_malloc_calls"] == a0["hip_malloc_calls"], (a0, a1)
    # the facade: CeresSolverBatch(pairs, devices) through the cached handle == the one-device call
    sys.path.insert(0, os.path.join(ROOT, "pnec_amd"))
    import pypnec
    lists = lambda a: [a[off[p]:off[p + 1]] for p in range(len(sizes))]
    T0 = []
    for p in range(len(sizes)):
        T = np.eye(4); T[:3, :3] = g.init_R[p].numpy(); T[:3, 3] = t0[p]; T0.append(T)
    one = pypnec.ceres_solver_batch(lists(f1), lists(f2), lists(cv), T0)
    two = pypnec.ceres_solver_batch(lists(f1), lists(f2), lists(cv), T0, devices=[0, 0])
    for a_, b_ in zip(one, two):
        np.testing.assert_array_equal(np.asarray(a_), np.asarray(b_))


def test_single_process_bench_line():
    """`bench.py --gpus 2 --single-process --share-gpu`: the multi-device handle driven by the bench (the comparison form
    for a SCALE run) prints the contract's line, every solve done, nothing allocated inside the timed steps."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--share-gpu",
                        "--pairs", "4000", "--steps", "3", "--warmup", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 1e5 and line["all_iterations_done"]
    assert line["config"]["hip_malloc_calls_during_timed_steps"] == 0 and line["config"]["shard_bounds"] == [0, 4000, 8000]


@pytest.mark.parametrize("chain", [False, True])
def test_both_launchers_print_the_same_records_digest_on_the_strong_scaling_workload(chain):
    """For the day an N-GPU box runs both: `bench.py --gpus N --workload kitti_all [--chain]` (one rank per GPU, one
    gather) and `... --single-process` (one process, the persistent multi-device handle) print `records_sha256`, the digest
    of the per-pair result records in pair order.  Here, with one GPU: the rank form with one rank, with two ranks sharing
    cuda:0 (gloo over pinned records), and the handle on the device list [0, 0] -- one and the same digest, whatever the
    sharding (RANSAC draws belong to the GLOBAL pair index)."""
    import json
    extra = ["--chain", "--in-flight", "1"] if chain else []
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "kitti_all", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"] + extra
    digests = {}
    for name, more in (("one rank", []), ("two ranks on cuda:0", ["--gpus", "2", "--share-gpu"]),
                       ("handle [0, 0]", ["--gpus", "2", "--share-gpu", "--single-process"])):
        r = subprocess.run(base + more, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert "error" not in line, (name, line)
        digests[name] = line["records_sha256"]
        if name == "two ranks on cuda:0":
            pr = line["per_rank"]
            assert len(pr["kernel_ms"]) == 2 and all(x and x > 0 for x in pr["kernel_ms"]) and sum(pr["pairs"]) == 23190
    assert len(set(digests.values())) == 1, digests
