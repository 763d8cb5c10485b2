#!/usr/bin/env python3
"""Analyse a PNEC_HIP_TRACE file (last launch): phase durations per workgroup and the overlap of
load phases between wavefronts sharing a CU / SIMD.  s_memtime ticks are converted with the clock
given as argv[3] (MHz, default 100 = the constant reference clock on gfx9-family parts)."""
import sys
from collections import defaultdict
import numpy as np
path, n_blocks = sys.argv[1], int(sys.argv[2])
mhz = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)[-n_blocks:]
t0, t1, t2 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2].astype(np.int64)
hw = (a[:, 3] & 0xFFFFFFFF).astype(np.int64)
xcc = (a[:, 3] >> 32).astype(np.int64) & 0xF
wave_id, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
ld, cp = (t1 - t0) / mhz, (t2 - t1) / mhz
print("ticks: load mean %.0f compute mean %.0f ; with %.0f MHz -> load %.2f us, compute %.2f us" %
      ((t1 - t0).mean(), (t2 - t1).mean(), mhz, ld.mean(), cp.mean()))
print("load p10/p50/p90 us: %.2f %.2f %.2f" % tuple(np.percentile(ld, [10, 50, 90])))
print("distinct xcc %d se %d sh %d cu %d simd %d wave_id %s" % (len(set(xcc)), len(set(se)), len(set(sh)), len(set(cu)), len(set(simd)), sorted(set(wave_id))))
groups = defaultdict(list)
for i in range(len(a)):
    groups[(xcc[i], se[i], sh[i], cu[i])].append(i)
print("CUs seen:", len(groups), " workgroups per CU: mean %.1f" % np.mean([len(v) for v in groups.values()]))
# overlap of load phases within a CU and within a SIMD
def overlap_stats(key_fn, name):
    g = defaultdict(list)
    for i in range(len(a)):
        g[key_fn(i)].append(i)
    conc = []
    for k, idx in list(g.items())[:64]:
        idx = np.array(idx)
        for i in idx[:200]:
            tm = (t0[i] + t1[i]) // 2
            conc.append(int(np.sum((t0[idx] <= tm) & (t1[idx] > tm))))
    print("%s: wavefronts in their load phase while another one loads: mean %.2f max %d" % (name, np.mean(conc), max(conc)))
overlap_stats(lambda i: (xcc[i], se[i], sh[i], cu[i]), "per CU")
overlap_stats(lambda i: (xcc[i], se[i], sh[i], cu[i], simd[i]), "per SIMD")
# timeline of one CU
k = sorted(groups)[0]
idx = np.array(groups[k]); o = np.argsort(t0[idx]); idx = idx[o][:24]
b = t0[idx].min()
print("first 24 workgroups of CU", k, ": (simd, wave_id, start, loaded, end) in us")
for i in idx:
    print("   ", int(simd[i]), int(wave_id[i]), "%.1f %.1f %.1f" % ((t0[i] - b) / mhz, (t1[i] - b) / mhz, (t2[i] - b) / mhz))
