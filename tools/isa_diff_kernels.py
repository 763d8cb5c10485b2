#!/usr/bin/env python3
"""Are two builds' kernels the SAME machine code?  Disassembles the gfx950 code objects of two .o / .so files and compares
every kernel they share, instruction by instruction (addresses and branch targets stripped).
  python tools/isa_diff_kernels.py A.o B.o [name-filter]  ->  one line per kernel: SAME / DIFF, instruction counts
Used for profiles/r06_asm_loads_ab.json: whether round 5's single-statement hand-written loads changed the headline kernel
at all (they are not called by it: it changed nothing -- the driver-timed -5 % of round 5 was the box)."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_asm_loads import OBJDUMP, disassemble  # noqa: E402


def kernels(path):
    out, cur = {}, None
    for l in disassemble(path, "gfx950", OBJDUMP):
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None or not l.strip():
            continue
        t = l.split("//")[0].strip()
        t = re.sub(r"<[^>]*>", "<>", t)
        if t and not t.endswith(":"):
            # branch offsets are relative: identical code has identical ones; keep the text as is
            out[cur].append(t)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = []
    for k in sorted(a):
        if k in b and flt in k and not k.startswith("__"):
            rows.append({"kernel": k, "same": a[k] == b[k], "instructions": [len(a[k]), len(b[k])]})
    for r in rows:
        print(("SAME " if r["same"] else "DIFF "), r["instructions"], r["kernel"][:140])
    return rows


if __name__ == "__main__":
    main()
