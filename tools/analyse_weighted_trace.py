#!/usr/bin/env python3
"""Reads the raw per-pair phase records of the weighted kernel (PNEC_HIP_TRACE_FRONT=<path>, [n_pairs, 12] uint64) and
prints: the launch's timeline (slots busy on average, when the last pairs start / end), and how well the launch-order
key (smallest / second eigenvalue of M at the first minimiser) predicts the long pairs.
   python tools/analyse_weighted_trace.py <path>"""
import sys, json
import numpy as np
h = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12).astype(np.float64)
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12)
tot, key = h[:, 6], h[:, 11] * 1e-12
st = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.float64); en = (raw[:, 7] >> np.uint64(32)).astype(np.float64)   # 100 MHz ticks
t0 = st.min(); t1 = en.max()
out = {"pairs": len(h), "launch_clocks": t1 - t0, "slots_busy_mean": (en - st).sum() / (t1 - t0), "last_start": (st.max() - t0) / (t1 - t0),
       "done_at": {q: (np.quantile(en, q) - t0) / (t1 - t0) for q in (0.5, 0.9, 0.99)},
       "total_clocks": {"p50": np.quantile(tot, .5), "p90": np.quantile(tot, .9), "p99": np.quantile(tot, .99), "max": tot.max()}}
long_ = tot > 3 * np.median(tot)
out["long_pairs"] = int(long_.sum())
out["long_pairs_share_of_work"] = tot[long_].sum() / tot.sum()
order = np.argsort(key)   # flattest first
for frac in (0.01, 0.02, 0.05, 0.1):
    top = order[: int(frac * len(h))]
    out[f"long_pairs_within_top_{frac}_by_key"] = int(long_[top].sum())
out["key_quantiles_long"] = [float(np.quantile(key[long_], q)) for q in (0.05, 0.5, 0.95)] if long_.any() else None
out["key_quantiles_rest"] = [float(np.quantile(key[~long_], q)) for q in (0.05, 0.5, 0.95, 0.99)]
# the ten slowest: start position in the launch and key rank
rank = np.empty(len(h)); rank[order] = np.arange(len(h))
out["slowest"] = [{"pair": int(p), "clocks": tot[p], "starts_at": (st[p] - t0) / (t1 - t0), "key": key[p], "key_rank": int(rank[p])} for p in np.argsort(-tot)[:10]]
print(json.dumps(out, indent=1, default=float))
