#!/usr/bin/env python3
"""profiles/rNN_full_pipeline_kernels.md from the outputs of tools/profile_pipeline.sh (+ optionally
tools/pmc_pipeline.sh's printed counters) and tools/bench_pipeline.py's JSON lines.
usage: summarize_pipeline_profile.py <prof_dir> <out.md> <pipeline.json> <pipeline_100k.json> [pmc.txt] [note]"""
import csv
import json
import os
import sys

prof, out_md, pj, pj100 = sys.argv[1:5]
pmc = sys.argv[5] if len(sys.argv) > 5 and os.path.exists(sys.argv[5]) else None
note = sys.argv[6] if len(sys.argv) > 6 else ""
R1 = {"ransac_eigensolver_kernel": 7532704, "weighted_eigensolver_kernel": 5858555, "select_kernel": 530836,
      "nec_eigensolver_kernel": 1026471, "lm_solve_kernel": 428824, "pack_kernel": 99952}
rows = [r for r in csv.DictReader(open(os.path.join(prof, "pipe_kernel_stats.csv"))) if ("pnec_hip" in r["Name"] or "anonymous" in r["Name"]) and "at::" not in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
p = json.load(open(pj))
p100 = json.load(open(pj100))
L = ["# rocprofv3 --kernel-trace --stats: full PNEC::Solve pipeline (tools/bench_pipeline.py 20000)", "",
     "20 000 pairs x 512 correspondences, 10 % gross outliers, reference-default Options; each stage is launched 4 times "
     "(1 warm-up + 3 timed).  " + note, "",
     "| kernel | calls | avg ns | min ns | max ns | round 1 avg ns |", "|---|---|---|---|---|---|"]
for r in rows:
    r1 = next((str(v) for k, v in R1.items() if k in r["Name"]), "")
    L.append("| `%s` | %s | %d | %s | %s | %s |" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r1))
g = p["gpu_ms"]
L += ["", "Stage wall times (host timers around each call, `%s`): " % os.path.basename(pj) +
      ", ".join("%s %.2f ms" % (k, v) for k, v in g.items()) +
      " -> **%.2f M full-pipeline pairs/s** (round 1: 1.39 M); at 100k pairs %.2f M (round 1: 1.67 M)." %
      (p["gpu_pairs_per_s_full_pipeline"] / 1e6, p100["gpu_pairs_per_s_full_pipeline"] / 1e6)]
if pmc:
    L += ["", "## SQ counters of the front-stage kernels (tools/pmc_pipeline.sh, separate --pmc passes; cycle counters in units of 4 clocks)", "", "```"]
    L += [l.rstrip() for l in open(pmc) if l.strip() and not l.startswith("[")]
    L += ["```"]
open(out_md, "w").write("\n".join(L) + "\n")
print("wrote", out_md)
