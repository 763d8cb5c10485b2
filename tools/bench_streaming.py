#!/usr/bin/env python3
"""BASELINE config 3 as the reference runs it: ~4.5k consecutive frame pairs, ONE CALL PER FRAME
(frame2frame.cc:122-141), through the streaming handle -- against the same sequence pre-batched.
Prints one JSON object.  Runs on the GPU box:  python tools/bench_streaming.py [pairs]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pnec_oracle as po
from pnec_amd import Batch, capi
from pnec_amd import simulation as sim
from pnec_amd.streaming import Stream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
offsets, f1, f2, c2, R_gt, t_gt, q0, t0 = sim.generate_kitti_like(P, mean_corr=500, seed=3)
f1, f2, c2, q0, t0 = (x.numpy() for x in (f1, f2, c2, q0, t0))
c9 = np.ascontiguousarray(np.transpose(c2, (0, 2, 1)).reshape(-1, 9))
pairs = [(f1[a:e], f2[a:e], c9[a:e]) for a, e in zip(offsets[:-1], offsets[1:])]
out = {"workload": f"KITTI-seq-00-like SYNTHETIC sequence, {P} ragged pairs ({int(np.diff(offsets).min())}..{int(np.diff(offsets).max())} "
                   f"correspondences), PNEC target-frame refinement, Ceres-default termination, host arrays in / pose out"}

# ---- batched: the whole sequence as one ragged batch, device-resident (what config 3 looked like in round 1)
dev = torch.device("cuda:0")
with Batch(capi.MODE_TARGET, offsets) as b:
    b.fill(torch.from_numpy(f1).to(dev), torch.from_numpy(f2).to(dev), torch.from_numpy(c2).to(dev))
    qd, td = torch.from_numpy(q0).to(dev), torch.from_numpy(t0).to(dev)
    res = None
    for _ in range(3):
        res = b.solve(qd, td, out=res)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter(); res = b.solve(qd, td, out=res); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    out["batched_pairs_per_s (inputs resident in HBM)"] = P / float(np.median(ts))
    ref_q = res.q.cpu().numpy()

# ---- streamed, pair by pair
def streamed(window):
    got = np.empty((P, 4))
    with Stream(max_corr=int(np.diff(offsets).max()), slots=max(window, 1)) as st:
        for p in range(64):                                    # warm-up
            st.wait(st.submit(capi.MODE_TARGET, *pairs[p], None, q0[p], t0[p]))
        t = time.perf_counter()
        tickets = []
        for p in range(P):
            tickets.append((p, st.submit(capi.MODE_TARGET, *pairs[p], None, q0[p], t0[p])))
            if len(tickets) >= window:
                i, tk = tickets.pop(0)
                got[i] = st.wait(tk).q[0]
        for i, tk in tickets:
            got[i] = st.wait(tk).q[0]
        dt = time.perf_counter() - t
    assert np.array_equal(got, ref_q), "streamed results differ from the batched call"
    return P / dt, dt / P * 1e6

for w in (1, 2, 4, 8):
    rate, us = streamed(w)
    out[f"streamed_window_{w}"] = {"pairs_per_s": rate, "us_per_pair": us, "bit_identical_to_batched": True}

# ---- the CPU oracle on the same per-frame pattern (one thread, like the reference)
t = time.perf_counter()
n_cpu = 300
for p in range(n_cpu):
    po.solve(po.MODE_TARGET, f1[offsets[p]:offsets[p + 1]], f2[offsets[p]:offsets[p + 1]], c2[offsets[p]:offsets[p + 1]], None,
             1e-13, q0[p], t0[p], po.default_options())
out["cpu_oracle_1_thread"] = {"pairs_per_s": n_cpu / (time.perf_counter() - t), "pairs": n_cpu}

# ---- the C++ facade's one-call latency (PNECCeres::Optimize, thread-local streaming handle)
for n in (100, 512):
    r = subprocess.run([os.path.join(ROOT, "pnec_amd", "pnec_host_demo"), str(n), "latency", "3000"], capture_output=True, text=True)
    out[f"facade_PNECCeres_Optimize_{n}_corr"] = json.loads(r.stdout.strip().splitlines()[-1])
for w in (1, 2, 4, 8):
    r = subprocess.run([os.path.join(ROOT, "pnec_amd", "pnec_host_demo"), "512", "stream", str(w), "20000"], capture_output=True, text=True)
    out[f"c_abi_stream_512_corr_window_{w}"] = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps(out), flush=True)
