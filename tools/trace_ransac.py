#!/usr/bin/env python3
"""What the long wavefronts of the RANSAC launch are made of (runs on the GPU box): the pipeline benchmark's batch, one
RANSAC call with PNEC_HIP_TRACE_FRONT=<path> (raw per-pair phase records), joined with the pairs' hypothesis counts.
usage: [PNEC_ES_SCHEME=0|1|2] trace_ransac.py [pairs] [out.json]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", "ransac_trace.json")
raw = os.path.join(root, "gpurun_out", "ransac_trace.bin")
os.environ["PNEC_HIP_TRACE_FRONT"] = raw
import numpy as np, torch
from pnec_amd import Batch, capi, simulation as sim
N = 512
dev = torch.device("cuda:0")
batch = Batch.uniform(capi.MODE_TARGET, B, N)
batch.set_eigensolver_scheme(int(os.environ.get("PNEC_ES_SCHEME", "0")))
qs = []
for c in range(0, B, 5000):
    m = min(5000, B - c)
    g = sim.generate(m, N, seed=1 + c, device=dev)
    bad = torch.rand(m, N, device=dev, generator=torch.Generator(device=dev).manual_seed(c)) < 0.10
    rnd = torch.randn(m, N, 3, dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(c + 1))
    rnd = rnd / rnd.norm(dim=-1, keepdim=True)
    g.bvs2 = torch.where(bad[..., None], rnd, g.bvs2)
    batch.fill(g.bvs1.reshape(-1, 3), g.bvs2.reshape(-1, 3), g.covs2.reshape(-1, 3, 3), first_pair=c, n_pairs=m)
    qs.append(g.init_q)
q0 = torch.cat(qs)
for _ in range(3):
    qr, tr, mask, cnt, its = batch.ransac_eigensolver(q0, seed=1)
torch.cuda.synchronize()
its = its.cpu().numpy(); cnt = cnt.cpu().numpy()
h = np.fromfile(raw, dtype=np.uint64).reshape(B, 12).astype(np.float64)
names = ["sample", "newton", "model", "score", "consume", "inliers", "start", "total", "its_sum", "trips", "rounds", "end"]
w = h[0::2]                                     # one record per wavefront (both pairs carry the wavefront's halves)
dur = (w[:, 11] - w[:, 6]) * 0.01               # us (10 ns ticks)
its2 = np.maximum(its[0::2], its[1::2]); itsum = its[0::2] + its[1::2]
order = np.argsort(dur)
def grp(sel):
    return {"waves": int(sel.sum()), "dur_us": float(dur[sel].mean()), "rounds": float(w[sel, 10].mean()), "trips": float(w[sel, 9].mean()),
            "its_max_of_two": float(its2[sel].mean()), "its_sum_of_two": float(itsum[sel].mean()),
            "clocks_per_pair": {n: float(w[sel, i].mean()) for i, n in enumerate(names) if n not in ("start", "end", "its_sum", "trips", "rounds")}}
res = {"pairs": B, "its_hist": np.bincount(np.minimum(its, 100)).tolist(),
       "all": grp(np.ones(len(w), bool)),
       "dur_p50_p90_p99_max_us": [float(np.percentile(dur, q)) for q in (50, 90, 99, 100)],
       "by_rounds": {str(int(r)): grp(w[:, 10] == r) for r in np.unique(w[:, 10])[:8]},
       "slowest_1pct": grp(dur >= np.percentile(dur, 99)),
       "trips_per_round_hist": None}
print(json.dumps(res))
json.dump(res, open(out, "w"), indent=1)

# ---- the launch's timeline (10 ns ticks -> us): when the last wavefront starts, when the launch has done 50 / 90 / 99 / 100 %
t0 = w[:, 6].min()
st_, en_ = (w[:, 6] - t0) * 0.01, (w[:, 11] - t0) * 0.01
tl = {"launch_us": float(en_.max()), "last_start_us": float(st_.max()), "done_50_90_99_us": [float(np.percentile(en_, q)) for q in (50, 90, 99)],
      "wavefront_time_over_2048_slots_us": float(dur.sum() / 2048.0), "slots_busy_on_average": float(dur.sum() / en_.max()),
      "wavefronts_ending_in_the_last_10pct_of_the_launch": int((en_ > 0.9 * en_.max()).sum())}
late = en_ > 0.9 * en_.max()
tl["those_wavefronts"] = {"dur_us": float(dur[late].mean()), "rounds": float(w[late, 10].mean()), "trips": float(w[late, 9].mean()),
                          "start_us": float(st_[late].mean())}
print(json.dumps(tl))
res["timeline"] = tl
json.dump(res, open(out, "w"), indent=1)
