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
