#!/usr/bin/env python3
"""Build-time backstop for the hand-written global loads (pnec_amd/csrc/pnec_device.hpp load_planes_saddr /
load_sets8x12_saddr, pnec_frontend.hip score_tiles_load / fib_prod_load4).

Since round 5 every such load sits in ONE inline-asm statement together with the s_waitcnt that retires it, so the
compiler has no place to put anything between issue and wait: correctness is by construction.  (Until then the loads
were issued by one statement and waited for by a later one, and the compiler did once spill a destination register in
between.)  This script keeps looking at what the compiler actually produced and fails the build if the rule is ever
broken again -- by a future edit that splits a statement, or by the compiler's own loads:

  for every scalar-base `global_load_dwordx2 vD, vO, s[..]` in the device code of libpnec_hip.so, no instruction between
  the load and the s_waitcnt vmcnt(n) that retires it may name vD.

vmcnt is modelled as the in-order FIFO it is on gfx9 (every vector-memory instruction enters it; `s_waitcnt
vmcnt(n)` leaves the youngest n outstanding).  The scan is linear through each function, which is conservative
across branches (a load stays in flight until a wait retires it).

usage: check_asm_loads.py [path/to/libpnec_hip.so] [--arch gfx950] [--objdump /path/to/llvm-objdump]
exit status 0 = clean, or nothing to check with (no llvm-objdump: the check is SKIPPED, the library stays)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")
RE_V = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
RE_LOAD = re.compile(r"^global_load_dwordx2 v\[(\d+):(\d+)\], v\d+, (s\[\d+:\d+\])")
RE_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def vregs(text):
    out = set()
    for m in RE_V.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_disassembly(lines):
    """returns (number of macro loads seen, list of violations)"""
    violations, seen = [], 0
    func = "?"
    fifo = []        # outstanding vector-memory instructions, oldest first: (is_macro_load, dest regs, text)
    prev = ""
    for raw in lines:
        line = raw.split("//")[0].strip()
        if not line:
            continue
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            func, fifo, prev = m.group(1), [], ""
            continue
        if line.endswith(":"):
            continue
        op = line.split()[0]
        # does this instruction touch a register a macro load in flight will write?
        used = vregs(line)
        ml = RE_LOAD.match(line)
        is_macro = bool(ml)
        for (mac, dest, text) in fifo:
            if mac and (used & dest):
                violations.append(f"{func}: `{line}` names v{sorted(used & dest)} while `{text}` is in flight")
        if op == "s_waitcnt" or op.startswith("s_waitcnt"):
            mv = RE_VMCNT.search(line)
            if mv:
                keep = int(mv.group(1))
                fifo = fifo[len(fifo) - keep:] if keep > 0 else []
        elif op.startswith(VMEM):
            dest = set(range(int(ml.group(1)), int(ml.group(2)) + 1)) if is_macro else set()
            seen += 1 if is_macro else 0
            fifo.append((is_macro, dest, line))
        elif op == "s_endpgm":
            fifo = []
        prev = line
    return seen, violations


def disassemble(so_path, arch="gfx950", objdump=OBJDUMP):
    OBJDUMP_ = objdump
    tmp = tempfile.mkdtemp(prefix="pnec_asmchk_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP_, "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if arch in f:
                r = subprocess.run([OBJDUMP_, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True)
                out.extend(r.stdout.split("\n"))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = sys.argv[1:]
    arch, objdump = "gfx950", OBJDUMP
    if "--arch" in args:
        i = args.index("--arch"); arch = args[i + 1]; del args[i:i + 2]
    if "--objdump" in args:
        i = args.index("--objdump"); objdump = args[i + 1]; del args[i:i + 2]
    so = args[0] if args else os.path.join(ROOT, "pnec_amd", "libpnec_hip.so")
    if not (os.path.isfile(objdump) and os.access(objdump, os.X_OK)):
        print(f"check_asm_loads: SKIPPED ({objdump} not found); the loads are single statements, correct by construction")
        return 0
    seen, bad = check_disassembly(disassemble(so, arch, objdump))
    print(f"check_asm_loads: {seen} scalar-base loads checked in {os.path.basename(so)} ({arch}), {len(bad)} violation(s)")
    for b in bad[:40]:
        print("  " + b)
    if seen == 0:
        print(f"  no scalar-base load found in the {arch} code object: nothing was checked (other target, or the fingerprint changed)")
        return 0 if arch != "gfx950" else 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
