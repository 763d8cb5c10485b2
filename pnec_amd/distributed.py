"""Multi-GPU: frame pairs are independent, so the batch shards with no data-path collective.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for the
tests).  Every rank solves a contiguous range of pairs; the only communication is ONE gather of
fixed-size result records (10 doubles = 80 B per solve) to rank 0 at the end -- a few MB even for
100k pairs per GPU, latency-bound, nowhere near the per-link xGMI ceiling.
The partition is a pure function of (sizes, world), so every rank knows every shard's size and no
size exchange is needed.

`RecordGather` is the pipelined form of that collective: the gather of step i runs on a side stream
(its own RCCL work queue) while the solve of step i+1 runs on the compute stream; result buffers are
double-buffered and guarded by events, so nothing is overwritten while the collective reads it.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

RECORD_WIDTH = 10  # q(4) t(3) cost iterations status


def partition(weights, world: int) -> np.ndarray:
    """Contiguous ranges of items balanced by weight (e.g. correspondences per pair).

    Returns bounds [world+1]; rank r owns items [bounds[r], bounds[r+1]).  Deterministic."""
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    bounds = np.zeros(world + 1, dtype=np.int64)
    if n == 0:
        return bounds
    c = np.concatenate([[0.0], np.cumsum(w)])
    total = c[-1]
    for r in range(1, world):
        target = total * r / world
        bounds[r] = int(np.searchsorted(c, target, side="left"))
    bounds[world] = n
    bounds = np.maximum.accumulate(np.minimum(bounds, n))
    return bounds


def partition_uniform(n_items: int, world: int) -> np.ndarray:
    return partition(np.ones(n_items), world)


def pack_records(res) -> torch.Tensor:
    """SolveResult (torch tensors) -> [S,10] float64 records."""
    return torch.cat([res.q, res.t, res.cost[:, None], res.iterations.to(torch.float64)[:, None],
                      res.status.to(torch.float64)[:, None]], dim=1)


def unpack_records(rec: torch.Tensor):
    from .batch import SolveResult
    return SolveResult(rec[:, 0:4], rec[:, 4:7], rec[:, 7], rec[:, 8].to(torch.int32),
                       rec[:, 9].to(torch.int32))


def gather_records(rec: torch.Tensor, world: int, rank: int, sizes=None, dst: int = 0):
    """ONE collective: gather every rank's records on `dst` (None elsewhere).

    sizes: records per rank when shards are ragged (known from `partition`); records are padded
    to the largest shard for the collective and trimmed on arrival."""
    if world == 1:
        return rec
    if sizes is None:
        sizes = [rec.shape[0]] * world
    m = int(max(sizes))
    if rec.shape[0] != sizes[rank]:
        raise ValueError("record count does not match the partition")
    if rec.shape[0] < m:
        pad = torch.zeros((m - rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        rec = torch.cat([rec, pad])
    rec = rec.contiguous()
    bufs = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    dist.gather(rec, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[: int(s)] for b, s in zip(bufs, sizes)])


class RecordGather:
    """The single gather of a step, issued off the solve's stream so that it overlaps the next step.

    Usage per step i:   slot = g.acquire()            # waits until the gather that last read this
                                                      # slot's buffers (step i-2) has finished
                        out = solve(..., out=bufs[slot])   # on the current (compute) stream
                        g.submit(slot, out)           # pack + gather on the side stream
    and once at the end g.drain() -> the last step's gathered records on `dst` (None elsewhere).
    On CPU tensors (gloo; the tests) the same calls run synchronously.

    Device forms (device = a CUDA device):
      * default: the collective itself takes the device records (backend "nccl" = RCCL over xGMI) and is
        enqueued on the side stream -- the production path, one GPU per rank;
      * host_staged=True: the side stream packs the records and copies them into a pinned host buffer; the
        collective (any backend that moves host tensors, i.e. gloo) runs when the slot comes round again or at
        drain().  Same events, same double buffering, same `sizes=` trimming -- it exists so that several ranks
        can share ONE GPU (RCCL refuses two ranks on one device), which is how this path is exercised on the
        single-GPU test boxes (`bench.py --share-gpu`, tests/test_distributed_gpu.py).
    force_collective=True issues the collective even for world == 1 (a one-rank RCCL communicator: the
    device call path of the production form, on a box with one GPU).
    """

    SLOTS = 2

    def __init__(self, world: int, rank: int, sizes=None, dst: int = 0, device=None, host_staged: bool = False,
                 force_collective: bool = False, slots: int = 2):
        self.world, self.rank, self.sizes, self.dst = world, rank, sizes, dst
        self.SLOTS = max(2, int(slots))   # result buffers in rotation (more than two: several steps in flight)
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.host_staged = bool(host_staged) and self.cuda
        self.force = bool(force_collective)
        self.next_slot = 0
        self.last = None
        self.collectives = 0      # collectives issued (bookkeeping for the tests)
        self.host_collective_s = 0.0   # host_staged: wall time this rank spent inside its collectives (incl. waiting for
        self.host_copy_wait_s = 0.0    # the slowest rank), and waiting for the records' copy into pinned memory
        if self.cuda:
            self.device = torch.device(device)
            self.side = torch.cuda.Stream(device=self.device)
            self.solved = [torch.cuda.Event() for _ in range(self.SLOTS)]
            self.gathered = [None] * self.SLOTS
            self.pinned = [None] * self.SLOTS     # host_staged: the slot's records on the host
            self.pending = [False] * self.SLOTS   # host_staged: copy enqueued, collective not yet run
            self.timing = []                      # (start, end) events around the side stream's pack + collective, last 64

    def _collective(self, rec: torch.Tensor):
        self.collectives += 1
        if self.world == 1 and self.force:
            bufs = [torch.empty_like(rec)]
            dist.gather(rec.contiguous(), bufs, dst=0)
            return bufs[0]
        return gather_records(rec, self.world, self.rank, self.sizes, self.dst)

    def _complete(self, slot: int) -> None:
        """host_staged: run the collective of a slot whose device-to-host copy was enqueued earlier."""
        if self.host_staged and self.pending[slot]:
            import time
            t0 = time.perf_counter()
            self.gathered[slot].synchronize()      # the copy into pinned memory has landed
            t1 = time.perf_counter()
            self.last = self._collective(self.pinned[slot])
            self.host_copy_wait_s += t1 - t0
            self.host_collective_s += time.perf_counter() - t1
            self.pending[slot] = False

    def acquire(self) -> int:
        slot = self.next_slot
        self.next_slot = (slot + 1) % self.SLOTS
        if self.cuda and self.gathered[slot] is not None:
            self._complete(slot)
            torch.cuda.current_stream(self.device).wait_event(self.gathered[slot])
        return slot

    def submit(self, slot: int, res) -> None:
        if not self.cuda:
            self.last = self._collective(pack_records(res))
            return
        self.solved[slot].record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.solved[slot])
            t_begin = torch.cuda.Event(enable_timing=True)
            t_begin.record(self.side)
            rec = pack_records(res)
            if self.host_staged:
                if self.pinned[slot] is None or self.pinned[slot].shape != rec.shape:
                    self.pinned[slot] = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
                self.pinned[slot].copy_(rec, non_blocking=True)
                self.pending[slot] = True
            else:
                self.last = self._collective(rec)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self.side)
            self.gathered[slot] = ev
            self.timing.append((t_begin, ev))
            if len(self.timing) > 64:
                del self.timing[0]

    def device_ms_per_collective(self):
        """Mean device time (ms) of the side stream's work per step -- pack the records + the collective (device form) or
        the copy into pinned memory (host_staged) -- over the last <= 64 submits; call after drain().  None on CPU."""
        if not self.cuda or not self.timing:
            return None
        self.side.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in self.timing]))

    def drain(self):
        if self.cuda:
            if self.host_staged:   # in submission order: the slot after the most recent one is the older
                for k in range(self.SLOTS):
                    self._complete((self.next_slot + k) % self.SLOTS)
            self.side.synchronize()
        return self.last
