"""world_size-2 gloo tests of the N>1 path: partition + the single gather of result records."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pnec_amd import distributed as pd


def test_partition_is_contiguous_balanced_and_deterministic():
    rng = np.random.default_rng(0)
    w = rng.integers(64, 700, size=4541)          # KITTI-00-like ragged pair sizes
    for world in (1, 2, 4, 8):
        b = pd.partition(w, world)
        assert b[0] == 0 and b[-1] == len(w) and (np.diff(b) >= 0).all()
        loads = [w[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(loads) - min(loads) <= 2 * w.max()
        np.testing.assert_array_equal(b, pd.partition(w, world))
    np.testing.assert_array_equal(pd.partition_uniform(10, 4), [0, 3, 5, 8, 10])
    np.testing.assert_array_equal(pd.partition([], 3), [0, 0, 0, 0])
    assert pd.partition_uniform(2, 8)[-1] == 2  # more ranks than items: some shards empty


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, sizes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = sizes[rank]
        base = sum(sizes[:rank])
        rec = torch.arange(base * pd.RECORD_WIDTH, (base + n) * pd.RECORD_WIDTH,
                           dtype=torch.float64).reshape(n, pd.RECORD_WIDTH)
        out = pd.gather_records(rec, world, rank, sizes=sizes)
        if rank == 0:
            total = sum(sizes)
            want = torch.arange(total * pd.RECORD_WIDTH, dtype=torch.float64).reshape(total, -1)
            q.put(bool(torch.equal(out, want)))
            res = pd.unpack_records(out)
            q.put(res.q.shape == (total, 4) and res.status.dtype == torch.int32)
        else:
            q.put(out is None)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [[5, 5], [7, 3], [0, 4]])
def test_single_gather_of_result_records_gloo(sizes):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


def test_pack_unpack_round_trip():
    from pnec_amd.batch import SolveResult
    S = 6
    res = SolveResult(torch.randn(S, 4, dtype=torch.float64), torch.randn(S, 3, dtype=torch.float64),
                      torch.rand(S, dtype=torch.float64), torch.arange(S, dtype=torch.int32),
                      torch.full((S,), 3, dtype=torch.int32))
    back = pd.unpack_records(pd.pack_records(res))
    assert torch.equal(back.q, res.q) and torch.equal(back.t, res.t)
    assert torch.equal(back.iterations, res.iterations) and torch.equal(back.status, res.status)


def test_c_abi_partition_is_the_python_partition():
    """pnec_hip_partition (the shard boundaries of pnec_hip_solve_pipeline_multi, one process, several GPUs) follows
    the rule of pnec_amd.distributed.partition (the multi-process bench): contiguous ranges of pairs balanced by
    correspondence count, a pure function of the sizes."""
    import ctypes as C
    from pnec_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(3)
    for counts in (rng.integers(1, 700, size=1000), np.full(64, 512), np.array([5, 0, 0, 900, 3]), np.array([7]),
                   rng.integers(0, 3, size=50)):
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        for parts in (1, 2, 3, 8, 16):
            got = np.zeros(parts + 1, dtype=np.int64)
            assert L.pnec_hip_partition(len(counts), off.ctypes.data, parts, got.ctypes.data) == 0
            np.testing.assert_array_equal(got, pd.partition(counts, parts))
    got = np.zeros(3, dtype=np.int64)
    assert L.pnec_hip_partition(0, None, 2, got.ctypes.data) == 0 and not got.any()
    assert L.pnec_hip_partition(4, None, 2, got.ctypes.data) == -1
    assert L.pnec_hip_partition(4, off.ctypes.data, 0, got.ctypes.data) == -1
    # the multi-device entry checks its arguments before touching a device
    assert L.pnec_hip_solve_pipeline_multi(0, None, 1, off.ctypes.data, None, None, None, None, None, None, None, None,
                                           None, None) == -1
