#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/profile_bench.sh) -> profiles/traffic_latest.json: per-launch HBM bytes of
the solver kernel from the separate FETCH_SIZE / WRITE_SIZE passes (gfx950 corrections of
MI355X_MICROARCH.md: FETCH_SIZE in KiB, x2 for coalesced streaming reads; WRITE_SIZE in KiB), the VALU
issue-busy fraction from the SQ pass, stamped with the identity of the device code they were measured on.
  python tools/make_traffic_json.py gpurun_out/prof_r02 [pairs corr iters]"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

src = sys.argv[1]
pairs, corr, iters = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (100_000, 512, 10)


def mean_counter(sub, name):
    vals = []
    f = os.path.join(src, sub, "bench_counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if "lm_solve_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name:
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_counter("fetch", "FETCH_SIZE")
write, nw = mean_counter("write", "WRITE_SIZE")
waves, _ = mean_counter("sq", "SQ_WAVES")
sq = {k: mean_counter("sq", k)[0] / waves for k in ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_SALU")}
line = None
for l in open(os.path.join(src, "stats.log")):
    if l.startswith("{") and '"roofline"' in l:
        line = json.loads(l)
launch = line["config"]["launch"]
out = {
    "source": f"{os.path.basename(src)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (separate passes, same bench command)",
    "kernel_sources_sha256": bench.kernel_sources_sha256(),
    "lib_sha256": hashlib.sha256(open(os.path.join(ROOT, "pnec_amd", "libpnec_hip.so"), "rb").read()).hexdigest(),
    "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "launches_averaged": [nf, nw],
    "correction": "FETCH_SIZE x1024 x2 (gfx950 reports half of a coalesced streaming read), WRITE_SIZE x1024",
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "workload": {"name": "sim100k", "pairs": pairs, "corr": corr, "iters": iters,
                 "geometry": [launch["corr_per_lane"], launch["waves_per_pair"], launch["lds_corr_per_lane"]]},
    "sq_counters_per_wave_quadcycles": sq,
    "valu_busy_frac": 2.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
    "valu_busy_note": "two wavefronts per SIMD: 2 x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES",
}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
