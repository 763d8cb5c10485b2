#!/usr/bin/env python3
"""Build-time guard for the hand-written global loads (PNEC_GLOBAL_LOAD_SADDR, pnec_amd/csrc/pnec_device.hpp).

Those loads are issued by one inline-asm statement and waited for by a later one, so the compiler believes the
destination registers are written when the issue statement returns.  It may then legally put a copy, a spill or
a re-use of such a register between the issue and the s_waitcnt -- an instruction that would read the register
before the load has landed, or be overwritten by it.  Nothing in the source can forbid that; this script looks at
what the compiler actually produced and fails the build if it ever happens:

  for every `s_mov_b64 sX, sY ; global_load_dwordx2 vD, vO, sX` pair in the gfx950 code of libpnec_hip.so (the macro's
  fingerprint), no instruction between the load and the s_waitcnt vmcnt(n) that retires it may name vD.

vmcnt is modelled as the in-order FIFO it is on gfx9 (every vector-memory instruction enters it; `s_waitcnt
vmcnt(n)` leaves the youngest n outstanding).  The scan is linear through each function, which is conservative
across branches (a load stays in flight until a wait retires it).

usage: check_asm_loads.py [path/to/libpnec_hip.so]      exit status 0 = clean
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
RE_SMOV = re.compile(r"^s_mov_b64 (s\[\d+:\d+\]), s\[\d+:\d+\]")
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
        is_macro = False
        if ml:
            ms = RE_SMOV.match(prev)
            is_macro = bool(ms) and ms.group(1) == ml.group(3)
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


def disassemble(so_path):
    tmp = tempfile.mkdtemp(prefix="pnec_asmchk_")
    try:
        local = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, cwd=tmp, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                r = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True)
                out.extend(r.stdout.split("\n"))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pnec_amd", "libpnec_hip.so")
    seen, bad = check_disassembly(disassemble(so))
    print(f"check_asm_loads: {seen} hand-written loads checked in {os.path.basename(so)}, {len(bad)} violation(s)")
    for b in bad[:40]:
        print("  " + b)
    if seen == 0:
        print("  no macro load recognised: the fingerprint has changed, fix this script")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
