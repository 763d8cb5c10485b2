#!/usr/bin/env python3
"""Static instruction mix of one solve-kernel instantiation (gfx950 ISA).

  python tools/isa_mix.py [--mode target] [--geom 8,1,3] [-D MACRO ...]

Compiles pnec_amd/csrc/pnec_solve_<mode>.hip to assembly, cuts out the kernel
lm_solve_kernel<MODE, CPL, WPP, LDSK, true> and prints, for the whole kernel and for its LM loop
(outermost loop), the number of VALU / SALU / LDS / memory instructions plus the most frequent
VALU opcodes; also the kernel's register/scratch footer.  The LDS-slot loop is counted once.
"""
import argparse, collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = {"nec": 0, "target": 1, "host": 2, "sym": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="target")
    ap.add_argument("--geom", default="8,1,3")
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--keep", default="/tmp/isa_mix.s")
    ap.add_argument("--region", default=None, help="with -D PNEC_ISA_MARKS: dump this region to stdout")
    args = ap.parse_args()
    cpl, wpp, ldsk = (int(x) for x in args.geom.split(","))
    src = os.path.join(ROOT, "pnec_amd/csrc", f"pnec_solve_{args.mode}.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-S", "--cuda-device-only", src, "-o", args.keep] + ["-D" + d for d in args.D]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    name = f"_ZN8pnec_hip15lm_solve_kernelILi{MODES[args.mode]}ELi{cpl}ELi{wpp}ELi{ldsk}ELb1ELi0EEEvNS_9SolveArgsE"
    lines = open(args.keep).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    body = lines[start:end]

    def mix(seg):
        c, ops = collections.Counter(), collections.Counter()
        for l in seg:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            if op.startswith("v_"):
                c["valu"] += 1
                ops[re.sub(r"_e(32|64)$", "", op) + ("_dpp" if " quad_perm" in l or " row_" in l else "")] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
                ops[op] += 1
            else:
                c["mem"] += 1
                ops[op] += 1
        return c, ops

    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and "=>This Loop Header" in l]
    print("kernel", name)
    c, _ = mix(body)
    print("  whole kernel:", dict(c))
    if heads:
        h = heads[-1]
        lab = body[h].split(":")[0]
        # the loop's blocks carry "in Loop: Header=<label>" annotations; it ends at the label after the last one
        tag = "Header=" + lab.lstrip(".L")
        last_in = max(i for i, l in enumerate(body) if tag in l)
        last = next((i for i in range(last_in + 1, len(body)) if re.match(r"\.LBB\d+_\d+:", body[i])), len(body) - 1) - 1
        c, ops = mix(body[h:last + 1])
        print(f"  LM loop ({lab}, {last - h} lines):", dict(c))
        for k, v in ops.most_common(24):
            print(f"    {v:5d} {k}")
    marks = [(i, l.split("PNEC_MARK", 1)[1].strip()) for i, l in enumerate(body) if "PNEC_MARK" in l]
    for (i, nm), (j, _) in zip(marks, marks[1:] + [(len(body), "")]):
        c, _o = mix(body[i:j])
        print(f"  region {nm:10s} lines {i:5d}..{j:5d}: {dict(c)}")
        if args.region and nm.strip('"') == args.region:
            print("\n".join(body[i:j]))
    for l in lines[end:end + 60]:
        if re.search(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs)", l):
            print("  " + l.strip("; ").strip())


if __name__ == "__main__":
    sys.exit(main())
