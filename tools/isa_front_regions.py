#!/usr/bin/env python3
"""Static FP64 instruction / flop count of the pieces of ONE evaluation of the eigenvalue function (es_value_grad in
pnec_frontend.hip): the file is compiled to assembly with -DPNEC_ISA_PROBE, which adds four tiny kernels -- the rotation
from the Cayley vector, M from the 36 sums, the whole evaluation with and without its gradient -- and their FP64
instructions are counted (v_fma / v_fmac = 2 flop, v_mul / v_add / v_rcp / v_rsq = 1).  This is the cross-check of
bench.py's flop model FLOP_ES_POINT = Cayley 47 + M 339 + eigenpair ~274 + gradient 292: the straight-line pieces must
come out at the model's numbers (the eigenpair part is a loop; its static count is one trip of each of its paths and is
printed for reference only); `--json` prints {"cayley":, "m":, "gradient":, "evaluation_static":} for the CPU test
(tests/test_bench_launch_cpu.py).   python tools/isa_front_regions.py [--json]"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOP = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_rcp_f64": 1, "v_rsq_f64": 1, "v_sqrt_f64": 1,
        "v_max_f64": 1, "v_min_f64": 1, "v_div_fmas_f64": 2, "v_div_fixup_f64": 1, "v_div_scale_f64": 1}


def count(out="/tmp/isa_front_probe.s"):
    src = os.path.join(ROOT, "pnec_amd/csrc/pnec_frontend.hip")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                    "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", "-DPNEC_ISA_PROBE", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    res = {}
    for key, want in (("cayley", "probe_cayley_kernel"), ("m", "probe_m_kernel"), ("with_gradient", "probe_value_grad_kernelILb1E"),
                      ("value_only", "probe_value_grad_kernelILb0E")):
        start = next(i for i, l in enumerate(lines) if want in l and not l.startswith("\t") and re.match(r"\S+:\s", l + " "))
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        c = collections.Counter()
        for l in lines[start:end]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = re.sub(r"_dpp$", "", re.sub(r"_e(32|64)$", "", l.split()[0]))
            if op in FLOP:
                c["fp64_instructions"] += 1
                c["flop"] += FLOP[op]
            if op.startswith("v_"):
                c["valu"] += 1
        res[key] = dict(c)
    return res


if __name__ == "__main__":
    r = count()
    summary = {"cayley": r["cayley"]["flop"], "m": r["m"]["flop"], "gradient": r["with_gradient"]["flop"] - r["value_only"]["flop"],
               "evaluation_static": r["with_gradient"]["flop"], "detail": r}
    if "--json" in sys.argv:
        print(json.dumps(summary))
    else:
        for k, v in r.items():
            print(f"{k:14s} {v}")
        print("flop: cayley", summary["cayley"], "| M from the 36 sums", summary["m"], "| gradient (with - without)", summary["gradient"],
              "| whole evaluation, static (eigenpair loop counted once per path)", summary["evaluation_static"])
