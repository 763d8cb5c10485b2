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
