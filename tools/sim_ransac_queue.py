#!/usr/bin/env python3
"""How a RANSAC round's eigenvalue minimisations pack onto the sixteen quads of a wavefront (CPU only).

The two-pair kernel hands a round's 32 minimisations (16 hypotheses of each pair) to sixteen quads through a queue; a
round's Newton phase is as long as the busiest quad.  This script takes the minimisations' lengths from the CPU checker
(pnec_oracle_es_last_trips: evaluations as the device's quad spends them) on the pipeline benchmark's data (512
correspondences, 10 % gross outliers) and replays scheduling rules over them:

  queue      the kernel's rule: quad q starts on task q, a finished quad takes the next task of the list
  lpt        the same with the list sorted longest first (perfect foresight: the bound for any ordering)
  lockstep K every task runs its first K trips in lockstep (two tasks per quad, one after the other), the rest is
             handed out longest-remaining first (perfect foresight of the rest: the bound for "rank by progress")
  pairs P    P pairs per queue (16 P tasks on sixteen quads)

Prints mean trips per pair for each rule.  Test tooling; nothing here is on a product path."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import pnec_oracle as po
from pnec_amd import simulation as sim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 240
N, SS, H = 512, 10, 16
L = po.lib()
L.pnec_oracle_es_last_trips.restype = C.c_int
g = sim.generate(P, N, seed=1)
gen = torch.Generator().manual_seed(0)
bad = torch.rand(P, N, generator=gen) < 0.10
rnd = torch.randn(P, N, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
rnd = rnd / rnd.norm(dim=-1, keepdim=True)
b2 = torch.where(bad[..., None], rnd, g.bvs2).numpy()
b1 = g.bvs1.numpy()
R0 = g.init_R.numpy()
bad = bad.numpy()

lengths = np.zeros((P, H), dtype=np.int64)   # trips of hypothesis h of pair p
f0 = np.zeros((P, H))                        # lambda_min at the start (the ordering key that was tried)
dirty = np.zeros((P, H), dtype=bool)         # sample holds a gross outlier
for p in range(P):
    v0 = po.rot_to_cayley(R0[p])
    for h in range(H):
        sel, draw = [], 0
        while len(sel) < SS:
            idx = min(int(L.pnec_oracle_rng_uniform(1, p, h, draw) * N), N - 1); draw += 1
            if idx not in sel: sel.append(idx)
        v = np.array([v0[c] + (L.pnec_oracle_rng_uniform(1, p, h, 1000 + c) - 0.5) * 0.02 for c in range(3)])
        Rs = po.cayley_to_rot(v)
        po.eigensolver(b1[p][sel], b2[p][sel], Rs)
        lengths[p, h] = L.pnec_oracle_es_last_trips()
        M = po.compose_m(b1[p][sel], b2[p][sel], Rs, skip_first=False)
        f0[p, h] = float(np.linalg.eigvalsh(M)[0])
        dirty[p, h] = bool(bad[p][sel].any())


def greedy(tasks, quads=16, head=0):
    """list scheduling: returns the makespan; `head` trips of every task have already been run"""
    free = np.zeros(quads, dtype=np.int64)
    for t in tasks:
        q = int(np.argmin(free))
        free[q] += max(int(t) - head, 0)
    return int(free.max())


out = {"pairs": P, "mean_trips_per_task": float(lengths.mean()), "p50": float(np.median(lengths)),
       "p99": float(np.percentile(lengths, 99)), "max": int(lengths.max()),
       "mean_clean": float(lengths[~dirty].mean()), "mean_dirty": float(lengths[dirty].mean()),
       "dirty_fraction": float(dirty.mean())}
for pairs_per_queue in (1, 2, 3, 4):
    groups = [np.concatenate([lengths[p + j] for j in range(pairs_per_queue)])
              for p in range(0, P - pairs_per_queue + 1, pairs_per_queue)]
    keys = [np.concatenate([f0[p + j] for j in range(pairs_per_queue)])
            for p in range(0, P - pairs_per_queue + 1, pairs_per_queue)]
    r = {"ideal": float(np.mean([max(np.ceil(t.sum() / 16), t.max()) for t in groups])) / pairs_per_queue,
         "queue": float(np.mean([greedy(t) for t in groups])) / pairs_per_queue,
         "lpt": float(np.mean([greedy(np.sort(t)[::-1]) for t in groups])) / pairs_per_queue,
         "by_start_value": float(np.mean([greedy(t[np.argsort(-k)]) for t, k in zip(groups, keys)])) / pairs_per_queue}
    for K in (2, 3, 4):
        per_quad = pairs_per_queue  # tasks each quad runs in lockstep
        r[f"lockstep{K}"] = float(np.mean([K * per_quad + greedy(np.sort(t)[::-1], head=K) for t in groups])) / pairs_per_queue
    out[f"{pairs_per_queue}_pairs_per_queue_trips_per_pair"] = r
print(json.dumps(out, indent=1))

# ---- the later rounds: how many pairs go on after their first sixteen hypotheses, and with how many
its = np.array([po.ransac_eigensolver(b1[p], b2[p], R0[p], seed=1, pair_id=p)[3] for p in range(P)])
later = its[its > 16] - 16
print(json.dumps({"ransac_iterations_mean": float(its.mean()), "pairs_beyond_one_round": float((its > 16).mean()),
                  "hypotheses_beyond_16_hist": np.bincount(np.minimum(later, 40)).tolist(),
                  "iterations_hist": np.bincount(np.minimum(its, 40)).tolist()}))
