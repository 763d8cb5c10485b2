#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ tree (tools/profile_bench.sh) into profiles/<name>.md:
kernel-trace stats for the solver kernels + per-launch PMC values with the gfx950 FETCH_SIZE
correction (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is in KiB and reports 1/2 of the bytes
of a coalesced streaming read on gfx950 -> x1024 x2; WRITE_SIZE x1024)."""
import csv
import os
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
out = [f"# rocprofv3 summary: {os.path.basename(dst)}", "", note, ""]

stats = os.path.join(src, "stats", "bench_kernel_stats.csv")
if os.path.exists(stats):
    out += ["## kernel trace (`rocprofv3 --kernel-trace --stats`), pnec_hip kernels", "",
            "| kernel | calls | avg ns | min ns | max ns | % of GPU time |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        if "pnec_hip" in r["Name"] or "pack_kernel" in r["Name"]:
            out.append(f"| `{r['Name']}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
    out.append("")
    out.append("(the rest of the trace is the torch kernels of the synthetic-data generator, outside the timed region)")
    out.append("")

# bench.py's own HIP-event timing printed during the traced run (must agree with the trace)
import json, re
log = os.path.join(src, "stats.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{") and '"roofline"' in line:
            d = json.loads(line)
            out += ["## bench.py line of the traced run (HIP events on the launch stream)", "",
                    f"* value = {d['value']:.4g} {d['unit']}, ms_per_step = {d['ms_per_step']:.3f}",
                    f"* roofline.kernel_ms (HIP events, mean of timed steps) = {d['roofline']['kernel_ms']:.3f} ms"
                    " -- compare with AverageNs above (the trace also includes the warm-up launch; profiled"
                    " runs clock a few % lower than un-profiled ones, MI355X_MICROARCH.md DVFS note)",
                    f"* roofline.achieved = {d['roofline']['achieved']:.1f} GB/s ({d['roofline']['frac']:.3f} of 8 TB/s),"
                    f" FP64 VALU {d['roofline']['valu']['achieved']:.1f} TFLOP/s ({d['roofline']['valu']['frac']:.3f} of 78.6)", ""]
pmc = defaultdict(list)
meta = {}
for sub in sorted(os.listdir(src)):
    f = os.path.join(src, sub, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        if "lm_solve_kernel" not in r["Kernel_Name"]:
            continue
        pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                  "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
if pmc:
    out += ["## PMC (separate `--pmc` passes, per launch of the solver kernel)", "",
            "launch: " + ", ".join(f"{k}={v}" for k, v in meta.items()), "",
            "| counter | launches | mean per launch | note |", "|---|---|---|---|"]
    for k, v in sorted(pmc.items()):
        m = sum(v) / len(v)
        n = ""
        if k == "FETCH_SIZE":
            n = f"KiB as reported; x1024 x2 (gfx950 correction) = {m * 1024 * 2 / 1e9:.3f} GB read from HBM per launch"
        if k == "WRITE_SIZE":
            n = f"KiB; x1024 = {m * 1024 / 1e6:.2f} MB written per launch"
        out.append(f"| {k} | {len(v)} | {m:.6g} | {n} |")
    out.append("")
    if "SQ_WAVES" in pmc and "SQ_INSTS_VALU" in pmc:
        w = sum(pmc["SQ_WAVES"]) / len(pmc["SQ_WAVES"])
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
                  "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS"):
            if k in pmc:
                out.append(f"* {k} per wave: {sum(pmc[k]) / len(pmc[k]) / w:.1f}")
        out.append("")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
